export RT_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 WORLD_SIZE=2
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 70 python -u -X faulthandler bench.py --gpus 2 --steps 2 --warmup 1 --tris 200000 --width 640 --height 360 > gpurun_out/two_$r.out 2> gpurun_out/two_$r.err &
done
wait
for r in 0 1; do echo "== rank $r"; tail -c 300 gpurun_out/two_$r.out; grep -v "amdgpu.ids\|hostname of the client" gpurun_out/two_$r.err | grep -A12 "most recent call first" | head -30; done
