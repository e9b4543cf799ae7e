cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys
j=json.load(open('$1'))
p=j['pcie_inclusive']
print('$2', 'both', round(p['value']/1e6,1), 'single', round(p['single_call']['value']/1e6,1), 'dist', round(p['distances_only']['value']/1e6,1))
"; }
python bench.py --no-cpu-baseline > /tmp/b1.json 2>/dev/null; show /tmp/b1.json no-cpu-baseline
python bench.py > /tmp/b2.json 2>/dev/null; show /tmp/b2.json with-cpu-baseline
OMP_WAIT_POLICY=passive python bench.py > /tmp/b3.json 2>/dev/null; show /tmp/b3.json with-cpu-baseline-passive
