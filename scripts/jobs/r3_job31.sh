for spec in "48000 20" "24000 40" "16000 60" "12000 80" "48000 200" "24000 400" "12000 800"; do set -- $spec
python bench.py --len $1 --steps $2 --warmup 5 --no-cpu --no-others --no-e2e --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); L=d['config']['samples_per_channel']; print('len',L,'steps',d['steps'],'ms_per_step %.4f'%d['ms_per_step'],' per 48000 samples: %.4f ms'%(d['ms_per_step']*48000/L), {k:round(v,3) for k,v in d['kernel_ms'].items()})"
done
