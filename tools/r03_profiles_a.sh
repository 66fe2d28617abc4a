# round-3 profile set, part A (on the GPU box): the headline and the synchronous form the plugin boundary uses
set -e
tools/profile_bench.sh r03_hrtf8192 > /dev/null && echo done hrtf8192
tools/profile_bench.sh r03_sync8192 --no-pipelined-mix > /dev/null && echo done sync8192
tools/profile_bench.sh r03_sync10240 --no-pipelined-mix --sources-per-gpu 10240 > /dev/null && echo done sync10240
tools/profile_bench.sh r03_sync65536 --no-pipelined-mix --sources-per-gpu 65536 > /dev/null && echo done sync65536
tools/profile_bench.sh r03_sync1M --no-pipelined-mix --sources-per-gpu 1048576 --marked-callbacks 32 > /dev/null && echo done sync1M
