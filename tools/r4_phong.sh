#!/bin/bash
# GPU box (round 4): Phong parity subset with both launch forms of march_phong2_kernel, then time + counters of tools/variants/phong2.py.  usage: tools/r4_phong.sh <tag>
TAG=${1:-r4p}
mkdir -p gpurun_out/$TAG
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "phong or c3_headline or render_matches or wild_fuzz" > gpurun_out/$TAG/pytest.log 2>&1; tail -3 gpurun_out/$TAG/pytest.log
VV_PHONG2=1 timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "phong or c3_headline or render_matches or wild_fuzz" > gpurun_out/$TAG/pytest_s1.log 2>&1; tail -3 gpurun_out/$TAG/pytest_s1.log
bash tools/traffic_split.sh ${TAG}_ts --phong --variants tools/variants/phong2.py > gpurun_out/$TAG/ts.log 2>&1
python3 - <<PY
lines=open('gpurun_out/${TAG}_ts/report.txt').read().splitlines()
hdr=[h.strip() for h in lines[0].split('|')]
for l in lines[1:]:
    c=[x.strip() for x in l.split('|')]
    d=dict(zip(hdr,c))
    print(c[0][:30].ljust(30), 'ms',d['ms'],'EA GB',c[4],'VALU',d['SQ_INSTS_VALU'],'VMEM',d['SQ_INSTS_VMEM_RD'],'WAIT_ANY',d['SQ_WAIT_ANY'],'WAVE_CYC',d['SQ_WAVE_CYCLES'])
PY
