#!/bin/bash
# round 6: the whole GPU suite on the in-tree build, then the headline with the dense mirror at 128^3 (default) / 256^3 / paged
cd "$(dirname "$0")/../.."
O=gpurun_out/r6suite; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-extra-legs --steps 400 --warmup 40"
for rep in 1 2; do
  $B > $O/c2_bits_default_$rep.json 2>$O/e.err
  ITM_MIRROR_BITS=8 $B > $O/c2_bits8_$rep.json 2>$O/e.err
  ITM_MIRROR_BITS=6 $B > $O/c2_bits6_$rep.json 2>$O/e.err
  ITM_MIRROR=paged $B > $O/c2_paged_$rep.json 2>$O/e.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r6suite/c2_*.json')):
    try:
        d = json.load(open(f)); r = d.get('roofline') or {}; a = d['config'].get('acceleration_structures') or {}
        print("%-28s %9.1f fps  %6.2f us/frame | raycast %6.2f us | mirror_bytes %d pages %d moves %d" % (f.split('/')[-1], d['value'], 1e3 * d['ms_per_step'], r.get('avg_kernel_us') or 0, a.get('mirror_bytes', -1), a.get('mirror_pages', -1), a.get('moves', -1)))
    except Exception as e:
        print(f, 'ERR', e)
PY
