# round-3 profile set, part B (on the GPU box): the other BASELINE configs and modes
set -e
tools/profile_bench.sh r03_unbatched_hrtf8192 --no-batched-launch > /dev/null && echo done unbatched
tools/profile_bench.sh r03_exactpeaks_hrtf8192 --exact-peaks > /dev/null && echo done exact
tools/profile_bench.sh r03_cfg3_hrtf4096 --workload hrtf4096 > /dev/null && echo done cfg3
tools/profile_bench.sh r03_cfg5_erhrtf4096 --workload erhrtf > /dev/null && echo done cfg5
tools/profile_bench.sh r03_cfg2_biquad256 --workload biquad > /dev/null && echo done cfg2
tools/profile_bench.sh r03_hrtf65536 --sources-per-gpu 65536 > /dev/null && echo done hrtf65536
tools/profile_bench.sh r03_biquad65536 --workload biquad --sources-per-gpu 65536 > /dev/null && echo done biquad65536
