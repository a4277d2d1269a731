R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04g; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for leg in "--no-roofline --no-f32-leg" "--no-roofline" "--no-f32-leg"; do
  tag=$(echo "$leg" | tr -d ' -')
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_$tag -o trace --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy $leg > $O/log_$tag.txt 2>&1
  python $R/tools/prof_csv_summary.py $O/p_$tag 8 > $O/stats_$tag.txt 2>&1; rm -rf $O/p_$tag
  echo "== $leg"; head -8 $O/stats_$tag.txt | cut -c1-140
done
