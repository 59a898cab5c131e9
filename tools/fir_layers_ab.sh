# blur-tail layer rows of bench.py for several builds: tools/fir_layers_ab.sh <variant> ...
for r in 1 2; do for v in "$@"; do
python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-side-configs --no-pcie-side --lib tools/bin/libmaua_$v.so 2>/dev/null | python -c "
import json,sys
p=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s r$r' % '$v', ' '.join('%.4f' % l['ms'] for l in p['layers'] if 'blur' in l['name']), '| family %.3f' % p['kernel_families']['upfirdn2d_tail']['ms_isolated'])"
done; done
