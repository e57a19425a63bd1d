# scripts/profile_round.sh + scripts/summarize_profile.py on the GPU box; only the summaries (gpurun_out/prof_<tag>_summary) and small raw files travel back.
#   bash scripts/run_profile_round.sh r05 ["m1b:full m1s:io c3:io c2:io c4:io"]
TAG=${1:-r05}
WL=${2:-"m1b:full m1s:io c3:io c2:io c4:io"}
cd /root/repo
export TFRA_BENCH_DETAIL_DIR=/tmp
bash scripts/profile_round.sh $TAG "$WL" > gpurun_out/prof_$TAG.log 2>&1
python scripts/summarize_profile.py $TAG gpurun_out/prof_${TAG}_summary > gpurun_out/prof_${TAG}_summary.log 2>&1
for W in m1b m1s c3 c2 c4; do
  # the bench line of the traced run itself = the stdout line that starts with {"metric" (rocprofv3 logs around it)
  if [ -f gpurun_out/prof_$TAG/${W}_bench_trace.log ]; then grep -a '^{"metric"' gpurun_out/prof_$TAG/${W}_bench_trace.log | tail -1 > gpurun_out/prof_${TAG}_summary/${TAG}_${W}_bench_line_under_rocprof.json; fi
done
du -sh gpurun_out/prof_$TAG | tail -1
find gpurun_out/prof_$TAG -name "*counter_collection.csv" -size +4M -delete
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +4M -delete
cat gpurun_out/prof_${TAG}_summary.log | cut -c1-700
