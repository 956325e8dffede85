set -e
o=gpurun_out/r03b
mkdir -p $o
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $o/$name.json 2> $o/$name.err || { echo "$name failed"; tail -3 $o/$name.err; exit 1; }; echo "$name done"; }
run bench_c4 --steps 4 --warmup 1
run bench_1m --tris 1000000 --steps 4 --warmup 1
run bench_c5 --workload closest_hit --tris 1000000 --steps 4 --warmup 1
run bench_c5_origins_inside --workload closest_hit --tris 1000000 --inflate 0.7 --steps 4 --warmup 1 --no-cpu
run bench_c3 --workload teaset --steps 4 --warmup 1
run bench_c3_1024spp_strong_n1 --workload teaset --spp 16 --total-passes 64 --gpus 1 --warmup 1 --no-cpu
run bench_c2 --workload cornell --width 1920 --height 1080 --spp 64 --steps 4 --warmup 1
run bench_c1_rtiow_640x480_128spp --workload rtiow --width 640 --height 480 --spp 128 --steps 4 --warmup 1
run bench_c4_device_build --builder device --steps 4 --warmup 1 --no-cpu
RTCORE_NODES=wide timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu > $o/bench_c4_wide_nodes.json 2> $o/bench_c4_wide_nodes.err
echo all done
