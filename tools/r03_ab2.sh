# same-box A/B of an environment switch: bash tools/r03_ab2.sh <tag> <VAR> <a> <b>
tag=$1; var=$2; a=$3; b=$4
for v in $a $b $a $b; do
  env $var=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-generation --no-loader 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], d['ms_per_step'], 'decoder alone', d.get('decoder_step',{}).get('alone_ms'), 'roofline', d['roofline']['frac'])" >> gpurun_out/${tag}_ab.txt
done
cat gpurun_out/${tag}_ab.txt
