for i in 1 2 3 4; do
  for cfg in "GPX_DIST2_LATE_COPYBACK=1" "GPX_DIST2_LATE_COPYBACK=0" "GPX_DIST2_LATE_COPYBACK=1 GPX_DIST_EARLY_BUF=0"; do
    r=$(env $cfg GPX_COMM=host GPX_FORCE_DEVICE=0 MASTER_ADDR=127.0.0.1 timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) tests/dist_worker.py --mode gpu2d --npts 1500 --mpts 777 --blk 256 2>&1 | grep -c "DIST_OK")
    echo "run $i [$cfg] ok=$r"
  done
done
