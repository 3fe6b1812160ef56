#!/bin/bash
set +e
mkdir -p gpurun_out
TELL_GEMM_S64=0 bash tools/profile_cmd.sh r05_c24_dec_s0 "decoder alone S64=0" python tools/decoder_profile.py faces_objects 32 20
TELL_GEMM_S64=1 SEQ_ANCHOR=bertadam_update bash tools/profile_cmd.sh r05_c24_dec_s1 "decoder alone S64=1" python tools/decoder_profile.py faces_objects 32 20
