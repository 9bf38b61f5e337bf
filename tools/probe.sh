#!/bin/bash
# developer probe: phase timeline of one gated layer launch + step time of the two schedules
timeout 300 python tools/gated_stamps.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/gated_probe.py 5 2>&1 | grep -v amdgpu.ids
