set -x
python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoders.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/ring_tests.log
for r in 2 3 4; do
  export TELL_GEMM_RING=$r
  python tools/bench_decoder_gemms.py > gpurun_out/ring${r}_decgemm.txt 2>&1
  python tools/bench_conv.py 32 > gpurun_out/ring${r}_conv.txt 2>&1
  python tools/decoder_profile.py faces_objects 32 20 > gpurun_out/ring${r}_decoder.txt 2>&1
  python tools/resnet_profile.py 32 20 > gpurun_out/ring${r}_resnet.txt 2>&1
done
for r in 2 4; do
  export TELL_GEMM_RING=$r
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/ring${r}_bench.json 2> gpurun_out/ring${r}_bench.err
done
tail -3 gpurun_out/ring_tests.log; tail -2 gpurun_out/ring*_decoder.txt gpurun_out/ring*_resnet.txt; cut -c1-200 gpurun_out/ring*_bench.json
