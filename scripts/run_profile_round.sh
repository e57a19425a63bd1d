# scripts/profile_round.sh + scripts/summarize_profile.py on the GPU box; only the summaries (gpurun_out/prof_<tag>_summary) and small raw files travel back.
cd /root/repo
bash scripts/profile_round.sh r04 > gpurun_out/prof_r04.log 2>&1
python scripts/summarize_profile.py r04 gpurun_out/prof_r04_summary > gpurun_out/prof_r04_summary.log 2>&1
for W in m1b c3 c2 c4; do tail -1 gpurun_out/prof_r04/${W}_bench_trace.log > gpurun_out/prof_r04_summary/r04_${W}_bench_line_under_rocprof.json; done
du -sh gpurun_out/prof_r04 | tail -1
find gpurun_out/prof_r04 -name "*counter_collection.csv" -size +4M -delete
find gpurun_out/prof_r04 -name "*kernel_trace.csv" -size +4M -delete
cat gpurun_out/prof_r04_summary.log | cut -c1-600
