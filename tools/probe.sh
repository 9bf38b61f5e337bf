#!/bin/bash
for d in 0 2 3 4 6 9; do echo "== chunk=$d"; CFX_GATE_DELAY=$d timeout 300 python tools/gated_probe.py 5 2>&1 | grep "gated launch\|peers"; done
