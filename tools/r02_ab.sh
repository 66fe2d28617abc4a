set -e
mkdir -p gpurun_out/r02
V=build/variants
tools/ab_libs.sh "" $V/libgas_prev.so - $V/libgas_prev.so - 
tools/ab_libs.sh "--no-pipelined-mix" $V/libgas_prev.so -
tools/ab_libs.sh "--exact-peaks --no-pipelined-mix" $V/libgas_prev.so -
tools/ab_libs.sh "--workload hrtf4096" $V/libgas_prev.so -
