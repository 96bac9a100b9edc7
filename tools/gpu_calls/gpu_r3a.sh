#!/bin/bash
# round 3, first GPU call: configs[3] / configs[4] for the first time (bench lines, kernel stats, at-size reference films)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3a
mkdir -p $O
export TMPDIR=/tmp
nproc > $O/nproc.txt
for w in sssdragon_bdpt cloud_bdpt; do
  timeout 900 python bench.py --workload $w --steps 4 --warmup 1 > $O/bench_$w.json 2> $O/bench_$w.err
  echo "bench $w rc=$?" >> $O/log.txt
  tail -c 600 $O/bench_$w.json >> $O/log.txt
done
for w in sssdragon_bdpt cloud_bdpt; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o $w -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_$w.json 2> $GRAFT_REPO_ROOT/$O/prof_$w.err )
  echo "prof $w rc=$?" >> $O/log.txt
  find /tmp/prof_$w -name "*kernel_stats.csv" -exec cp {} $O/${w}_kernel_stats.csv \;
done
timeout 1500 python3 oracle/gen_golden_1080p.py sssdragon sssdragon_asis cloud cloud_asis full_asis > $O/golden.log 2>&1
echo "golden rc=$?" >> $O/log.txt
timeout 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err
echo "bench full rc=$?" >> $O/log.txt
cat $O/log.txt
