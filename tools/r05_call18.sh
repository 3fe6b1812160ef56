#!/bin/bash
# decode-step timelines at 128 captions per batch: greedy (128 rows, weight-streaming step) and beam 4 (512 rows, layer by layer)
set +e
mkdir -p gpurun_out
SEQ_ANCHOR=greedy_update bash tools/profile_cmd.sh r05_generate_b128 "greedy generation, B=128: python bench.py --generate --batch 128 --beam 1 --steps 1 --warmup 1 --gen-serial" python bench.py --generate --batch 128 --beam 1 --steps 1 --warmup 1 --gen-serial
SEQ_ANCHOR=beam_update bash tools/profile_cmd.sh r05_beam_b128 "beam-4 generation, B=128: python bench.py --generate --batch 128 --beam 4 --steps 1 --warmup 1 --gen-serial" python bench.py --generate --batch 128 --beam 4 --steps 1 --warmup 1 --gen-serial
