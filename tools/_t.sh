set -e
echo "== default"; timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2
echo "== grid nodes forced"; PT_GRID_NODES=1 timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
for g in 0 1; do echo "== PT_GRID_NODES=$g"; PT_GRID_NODES=$g timeout -k 10 300 python tools/bench_configs.py "tree in HBM" "scene 2" C5 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'][:34], d['kernel'], d['kernel_ms'], d['Msamples/s'], d['node_visits/seg'])"; done
