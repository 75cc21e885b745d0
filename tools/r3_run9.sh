for cfg in "5" "3 --dtype bf16"; do
  timeout -k 10 280 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --breakdown 2>&1 >/dev/null | grep -v amdgpu | cut -c1-100
done
