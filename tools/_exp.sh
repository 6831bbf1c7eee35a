timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline --stage-times --steps 4 --warmup 2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['stage_ms_per_step'].items()}, d['config']['classified_ok'])"
