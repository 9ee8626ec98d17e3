# same-box A/B of two builds of the library (ligero_amd/lib/ab_<name>.so, e.g. make BUILD=../../build/x LIB=../lib/ab_plain.so HIPFLAGS="... -DLG_PLAIN_STORES")
for rep in 1 2; do for v in nt plain; do for w in poseidon s20 s22; do
  st=20; [ $w = poseidon ] && st=200
  LIGERO_HIP_LIB=$PWD/ligero_amd/lib/ab_$v.so python bench.py --workload $w --steps $st --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$w', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['stage_ms'].items() if k!='samples'})"
done; done; done
