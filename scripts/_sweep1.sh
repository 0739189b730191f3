source scripts/paced_sweep.sh true
RANKS=0,1,2,4 run default X=1
for agg in 1 2 8; do run agg$agg GPX_DIST_AGG=$agg; done
for b in bulk chunks main; do run bulk_$b GPX_DIST_BULK=$b; done
run agg1_bulk GPX_DIST_AGG=1 GPX_DIST_BULK=bulk
run agg2_bulk GPX_DIST_AGG=2 GPX_DIST_BULK=bulk
run small128 GPX_DIST2_SMALL_MAX=0
