#!/bin/bash
# Round-6 evidence in one GPU-box call (repo root): bash tools/r06_profiles.sh  -> gpurun_out/r06_*
# (copied into profiles/ by hand afterwards; every file names the command that produced it)
set +e
P="--windows 1 --burn-in 0"
bash tools/profile_bench.sh r06 --steps 9 --warmup 1 $P --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest
bash tools/profile_bench.sh r06_serial --serial --steps 4 --warmup 1 $P --no-cpu-baseline --no-secondary --no-dp-selftest
SEQ_ANCHOR=bertadam bash tools/profile_cmd.sh r06_decoder "decoder half of the step alone (fwd + loss + bwd + BertAdam), B=32, step graph: python tools/decoder_profile.py faces_objects 32 20" python tools/decoder_profile.py faces_objects 32 20
bash tools/profile_cmd.sh r06_resnet "ResNet-152 alone, B=32, train-mode BatchNorm: python tools/resnet_profile.py 32 20 train" python tools/resnet_profile.py 32 20 train
bash tools/profile_cmd.sh r06_resnet_eval "ResNet-152 alone, B=32, eval (BatchNorm folded): python tools/resnet_profile.py 32 20 eval" python tools/resnet_profile.py 32 20 eval
SEQ_ANCHOR=greedy_update bash tools/profile_cmd.sh r06_generate "greedy generation, B=32: python bench.py --generate --beam 1 --steps 1 --warmup 1" python bench.py --generate --beam 1 --steps 1 --warmup 1
SEQ_ANCHOR=beam_update bash tools/profile_cmd.sh r06_beam "beam-4 generation, B=32: python bench.py --generate --beam 4 --steps 1 --warmup 1" python bench.py --generate --beam 4 --steps 1 --warmup 1
SEQ_ANCHOR=beam_update bash tools/profile_cmd.sh r06_beam_b128 "beam-4 generation, 128 captions per batch (512 rows, layer by layer): python bench.py --generate --beam 4 --batch 128 --steps 1 --warmup 1" python bench.py --generate --beam 4 --batch 128 --steps 1 --warmup 1
SEQ_ANCHOR=greedy_update bash tools/profile_cmd.sh r06_generate_b128 "greedy generation, 128 captions per batch: python bench.py --generate --beam 1 --batch 128 --steps 1 --warmup 1" python bench.py --generate --beam 1 --batch 128 --steps 1 --warmup 1
bash tools/pmc_traffic.sh gemm_nt_q4 gpurun_out/r06_pmc_gemm_traffic.json "gemm_nt_q4_kernel<bf16,256,256>" > /dev/null
bash tools/pmc_generate_traffic.sh 1 gpurun_out/r06_pmc_generate_greedy_traffic.json > /dev/null 2>&1
bash tools/pmc_generate_traffic.sh 4 gpurun_out/r06_pmc_generate_beam4_traffic.json > /dev/null 2>&1
python tools/bench_skinny.py 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_skinny_bench.txt
python tools/bench_skinny.py 128 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_skinny_bench.txt
python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_attention.txt
python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_roberta_gemms.txt
python tools/bench_dynconv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_dynconv.txt
python tools/bench_decoder_gemms.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_decoder_gemms.txt
for b in 1 4; do python bench.py --generate --beam $b 2>/dev/null | tail -1 > gpurun_out/r06_generate_beam$b.json; done
(time python bench.py 2> gpurun_out/r06_bench.err | tail -1 > gpurun_out/r06_bench.json) 2> gpurun_out/r06_bench_time.txt
