# round-2 profile set, part A (on the GPU box): headline, ordered, exact peaks, cfg3
set -e
tools/profile_bench.sh r02_hrtf8192 > /dev/null && echo done hrtf8192
tools/profile_bench.sh r02_unbatched_hrtf8192 --no-batched-launch > /dev/null && echo done unbatched
tools/profile_bench.sh r02_ordered_hrtf8192 --no-pipelined-mix > /dev/null && echo done ordered
tools/profile_bench.sh r02_exactpeaks_hrtf8192 --exact-peaks > /dev/null && echo done exact
tools/profile_bench.sh r02_cfg3_hrtf4096 --workload hrtf4096 > /dev/null && echo done cfg3
