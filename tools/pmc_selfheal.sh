#!/bin/bash
# the pipeline under a tool that serialises the dispatches (rocprofv3 --pmc) WITHOUT LVT_AMD_ORDERING=events: the first gate time-out moves the handle to event ordering
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sh
s=$(date +%s.%N)
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_sh -o bench -- python /root/repo/bench.py --steps 60 --warmup 5 --skip kernels,sync,batch,lists_ab,configs,cpu > /tmp/pmc_sh.json 2> /tmp/pmc_sh.err
e=$(date +%s.%N)
echo "rc=$? wall $(python -c "print(round($e-$s,1))") s"
python - <<'PY'
import json
r = json.loads(open("/tmp/pmc_sh.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms_per_step", r["ms_per_step"], "not tracking", r["tracking"]["frames_not_tracking"], "| error:", r["tracking_error_string"][:200], "| se3", r.get("se3", {}).get("pass"))
PY
