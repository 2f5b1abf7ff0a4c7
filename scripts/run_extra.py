"""One of bench.py's `extra` workloads on its own (dev tool; profile with scripts/prof_cmd.sh):
    python scripts/run_extra.py em_custom|em_hmmlearn|decode_custom|pipe39|decode|refmfcc [n_utts]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from sapr_amd.frontend import BENCH, MfccPlan  # noqa: E402

what = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = torch.device("cuda", 0)
pcm = bench.synth_pcm(torch, n, seed=1234, device=dev)
lens = np.full(n, bench.N_SAMP, dtype=np.int64)
feats, _ = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)(pcm, lens)
if what == "em_custom":
    out = bench.extra_em_custom(torch, dev, feats, n)
elif what == "em_hmmlearn":
    out = bench.extra_em_hmmlearn(torch, dev, feats, n)
elif what == "decode_custom":
    out = bench.extra_decode_custom(torch, dev, feats, n)
elif what == "pipe39":
    out = bench.extra_pipeline39(torch, dev, pcm, n)
elif what == "decode":
    m = min(n, 2200)
    models = bench.build_models(feats[: m * bench.T_FRAMES].cpu().numpy().reshape(m, bench.T_FRAMES, bench.D))
    out = bench.extra_decode_sensitivity(torch, dev, feats, n, models)
elif what == "refmfcc":
    out = bench.extra_mfcc_reference_preset(torch, dev, min(n, 10000))
else:
    raise SystemExit(what)
print(json.dumps(out, indent=1))
