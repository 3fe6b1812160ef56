# round-3 A/B driver: bash tools/r03_ab.sh <tag> [pytest args] -- runs tests + profiles, logs under gpurun_out/<tag>_*
tag=${1:-ab}; shift
python -m pytest ${*:-tests} -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/${tag}_tests.log
python tools/decoder_profile.py faces_objects 32 20 > gpurun_out/${tag}_decoder.txt 2>&1
python tools/decoder_profile.py flattened 16 20 > gpurun_out/${tag}_decoder2.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -4 gpurun_out/${tag}_tests.log; grep median gpurun_out/${tag}_decoder*.txt; tail -3 gpurun_out/${tag}_bench.err; cut -c1-220 gpurun_out/${tag}_bench.json
