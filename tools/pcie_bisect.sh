show() { python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); p=d.get('pcie_inclusive', d); print('$1', p['pinned'], d.get('value'))"; }
timeout -k 10 200 python3 bench.py --pcie-only 2>/dev/null | show only || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-hamming --no-config5 --steps 5 2>/dev/null | show ext+pcie || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-hamming --no-config5 --steps 5 --frames-resident 256 2>/dev/null | show ext256+pcie || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-config5 --steps 5 2>/dev/null | show ext+ham+pcie || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-hamming --steps 5 2>/dev/null | show ext+c5+pcie || exit 1
