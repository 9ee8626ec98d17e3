set -u
i=0
for cfg in "24 64 61 1" "344 128 62 1" "40 1024 63 3" "20 8192 64 1" "344 128 65 16" "8 4096 66 1"; do
  i=$((i+1))
  LG_FORCE_CHUNKS=$((i % 2 * 2)) timeout -k 5 90 python tools/fuzz_api_sequences.py 25 $cfg 2>&1 | grep "^api sequence\|Error\|assert" | cut -c1-400
done
timeout -k 10 300 python tools/soak_sharded.py 100 8 10036 4096 2>&1 | grep "^soak\|Error"
timeout -k 10 300 python tools/soak_prover.py 60 3 2>&1 | grep "^soak\|Error"
timeout -k 10 300 python tools/soak_sharded_prover.py 40 4 s18 2>&1 | grep "^soak\|Error"
