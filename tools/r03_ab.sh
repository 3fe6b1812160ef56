# round-3 A/B driver: bash tools/r03_ab.sh <tag> -- runs tests + profiles, logs under gpurun_out/<tag>_*
tag=${1:-ab}
python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoders.py -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/${tag}_tests.log
for f in 0 1; do
  TELL_BN_FUSE=$f python tools/resnet_profile.py 32 20 train > gpurun_out/${tag}_resnet_fuse$f.txt 2>&1
done
python tools/resnet_profile.py 32 20 eval > gpurun_out/${tag}_resnet_eval.txt 2>&1
python tools/decoder_profile.py faces_objects 32 20 > gpurun_out/${tag}_decoder.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -4 gpurun_out/${tag}_tests.log; cat gpurun_out/${tag}_resnet_*.txt | grep median; grep median gpurun_out/${tag}_decoder.txt; cut -c1-220 gpurun_out/${tag}_bench.json
