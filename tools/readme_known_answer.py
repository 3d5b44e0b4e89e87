#!/usr/bin/env python3
"""Writes the envelopes of the README known answer (tests/readme_known_answer.py, tests/test_readme_known_answer.py):
`--cpu` the float64 oracle's -> gpurun_out/r04_readme_known_answer_oracle.json (container or box),
`--gpu` oracle + HIP path on the MI355X -> gpurun_out/r04_readme_known_answer.json (copy to profiles/).
TEST INFRASTRUCTURE: imports oracle/."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import readme_known_answer as rk  # noqa: E402
from tests import test_readme_known_answer as t  # noqa: E402

HEAD = {"source": "/root/reference/README.md:258-273 (values), /root/reference/examples/probability_demo.py:45-86 (protocol)",
        "seeds": rk.N_SEEDS, "readings": [r[0] for r in rk.READINGS], "consistent_readings": list(rk.CONSISTENT),
        "rule": "README value v is inside when min - 0.0005 <= v <= max + 0.0005 (printed with %.3f); `misses` are 1-based print positions"}

if __name__ == "__main__":
    if "--gpu" in sys.argv:
        res = dict(HEAD, oracle_f64=t.oracle_envelopes(), hip=t.hip_envelopes())
        from cutter_vad_amd.engine import Engine
        with Engine(t._blob(), model_version=5, max_streams=8) as e:
            res["device"] = e.info()["device_name"]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        path = os.path.join(ROOT, "gpurun_out", "r04_readme_known_answer.json")
    else:
        res = dict(HEAD, oracle_f64=t.oracle_envelopes())
        path = os.path.join(ROOT, "gpurun_out", "r04_readme_known_answer_oracle.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print({k: v["misses"] for k, v in res["oracle_f64"].items()}, {k: v["misses"] for k, v in res.get("hip", {}).items()})
