#!/bin/bash
# Round-5 evidence in one GPU-box call (repo root): bash tools/r05_profiles.sh  -> gpurun_out/r05_*
# (copied into profiles/ by hand afterwards; every file names the command that produced it)
set +e
bash tools/profile_bench.sh r05 --steps 9 --warmup 1 --no-cpu-baseline --no-secondary --no-generation --no-loader
bash tools/profile_bench.sh r05_serial --serial --steps 4 --warmup 1 --no-cpu-baseline --no-secondary
SEQ_ANCHOR=bertadam bash tools/profile_cmd.sh r05_decoder "decoder half of the step alone (fwd + loss + bwd + BertAdam), B=32, step graph: python tools/decoder_profile.py faces_objects 32 20" python tools/decoder_profile.py faces_objects 32 20
bash tools/profile_cmd.sh r05_resnet "ResNet-152 alone, B=32, train-mode BatchNorm: python tools/resnet_profile.py 32 20 train" python tools/resnet_profile.py 32 20 train
bash tools/profile_cmd.sh r05_resnet_eval "ResNet-152 alone, B=32, eval (BatchNorm folded): python tools/resnet_profile.py 32 20 eval" python tools/resnet_profile.py 32 20 eval
SEQ_ANCHOR=greedy_update bash tools/profile_cmd.sh r05_generate "greedy generation, B=32: python bench.py --generate --beam 1 --steps 1 --warmup 1" python bench.py --generate --beam 1 --steps 1 --warmup 1
SEQ_ANCHOR=beam_update bash tools/profile_cmd.sh r05_beam "beam-4 generation, B=32: python bench.py --generate --beam 4 --steps 1 --warmup 1" python bench.py --generate --beam 4 --steps 1 --warmup 1
bash tools/pmc_traffic.sh gemm_nt_q4 gpurun_out/r05_pmc_gemm_traffic.json "gemm_nt_q4_kernel<bf16,256,256>" > /dev/null
bash tools/pmc_kernel.sh attn_self gpurun_out/r05_pmc_attention.txt SQ_BUSY_CYCLES,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_INST_CYCLES_VMEM,SQ_VALU_MFMA_BUSY_CYCLES -- python tools/bench_attention.py > /dev/null
python tools/bench_attention.py > gpurun_out/r05_attention.txt 2>&1
python tools/bench_skinny.py 32 > gpurun_out/r05_skinny_bench.txt 2>&1
python tools/bench_skinny.py 128 >> gpurun_out/r05_skinny_bench.txt 2>&1
python tools/bench_roberta_gemms.py > gpurun_out/r05_roberta_gemms.txt 2>&1
for b in 1 4; do python bench.py --generate --beam $b 2>/dev/null | tail -1 > gpurun_out/r05_generate_beam$b.json; done
python bench.py 2> gpurun_out/r05_bench.err | tail -1 > gpurun_out/r05_bench.json
python tools/bench_dynconv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_dynconv.txt
bash tools/pmc_kernel.sh gemm_nt_q4 gpurun_out/r05_pmc_gemm_mfma.txt SQ_BUSY_CYCLES,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_INST_CYCLES_VMEM,SQ_VALU_MFMA_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_ANY -- python tools/bench_roberta_gemms.py > /dev/null
bash tools/pmc_generate_traffic.sh 1 gpurun_out/r05_pmc_generate_greedy_traffic.json > /dev/null 2>&1
bash tools/pmc_generate_traffic.sh 4 gpurun_out/r05_pmc_generate_beam4_traffic.json > /dev/null 2>&1
python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_conv_default.txt
python tools/bench_layernorm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_layernorm.txt
python tools/bench_decoder_gemms.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_decoder_gemms.txt
