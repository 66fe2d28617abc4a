# round-2 profile set, part B: cfg2, cfg5, 65536-source runs
set -e
tools/profile_bench.sh r02_cfg2_biquad256 --workload biquad > /dev/null && echo done cfg2
tools/profile_bench.sh r02_cfg5_erhrtf4096 --workload erhrtf > /dev/null && echo done cfg5
tools/profile_bench.sh r02_hrtf65536 --sources-per-gpu 65536 > /dev/null && echo done hrtf65536
tools/profile_bench.sh r02_biquad65536 --workload biquad --sources-per-gpu 65536 > /dev/null && echo done biquad65536
tools/profile_bench.sh r02_xcddirs_hrtf8192 --xcd-directions > /dev/null && echo done xcddirs
tools/profile_bench.sh r02_xcdorder_hrtf65536 --sources-per-gpu 65536 --xcd-order > /dev/null && echo done xcdorder65536
