fail=0
for seed in 21 22 23 24 25 26 27 28 29 30 31 32; do
  for cfg in "2 1500 256" "4 2100 256" "3 1900 128"; do
    set -- $cfg
    r=$(GPX_COMM=host GPX_FORCE_DEVICE=0 MASTER_ADDR=127.0.0.1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) tests/dist_worker.py --mode gpu2d-chaos --npts $2 --mpts 777 --blk $3 --chaos $seed 2>&1 | grep -c "DIST_OK")
    if [ "$r" != "1" ]; then echo "FAIL seed $seed world $1 n $2 nb $3"; fail=1; fi
  done
done
echo "chaos sweep done fail=$fail"
