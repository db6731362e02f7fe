for a in "--streams 3 --sub-batch 128" "--streams 4 --sub-batch 128" "--streams 2 --sub-batch 128" "--streams 3 --sub-batch 256 --batch 768" "--streams 3 --sub-batch 64" "--streams 6 --sub-batch 64 --batch 384" "--streams 3 --sub-batch 128"; do
python bench.py --config 2 --single-config --steps 20 --warmup 4 --cpu-seconds 0 --e2e-stars 0 --no-kernel-timing $a 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$a', round(d['value']), d['config'].get('stars_per_step'))"
done
