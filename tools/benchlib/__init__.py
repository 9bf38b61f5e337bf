"""bench.py in four reviewable parts: workload.py (a: the workload and its state), schedules.py (b: the plan builders), safety.py (c: validate /
fall back / states_consistent - the N > 1 safety net, unit-tested on CPU), report.py (d: the JSON line)."""
