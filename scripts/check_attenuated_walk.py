"""attenuated mask: the carried-block kernel against the per-ping kernel that takes both medians of every ping from
memory (reached with S % 4 != 0: a padding sample below the layer), on a volume with attenuated pings, NaN, ragged
layer limits (development aid)."""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, ".")
from echopype_amd import ops

rng = np.random.default_rng(3)
C, P, S = 2, 3000, 600
sv = -70 + 4 * rng.standard_normal((C, P, S)) - 10 * np.linspace(0, 1, S)
att = rng.random((C, P)) < 0.05
sv[att] -= rng.uniform(3, 30, size=att.sum())[:, None]
sv[rng.random((C, P, S)) < 0.03] = np.nan
sv[0, 500:520] = np.nan
sv[1, 900:960, 100:400] = -71.25        # flat: more than 64 values in the median's bin
depth = np.broadcast_to(1.0 + 0.5 * np.arange(S), (C, P, S)).copy()
depth[1, 1500:] *= 1.1                  # layer limits change once
depth[0, 2000:2010] *= 1 + 0.05 * rng.random((10, 1))   # ... and from ping to ping
for dt in (torch.float64, torch.float32):
    svt, rgt = torch.from_numpy(sv).to(dt).cuda(), torch.from_numpy(depth).to(dt).cuda()
    new = ops.attenuated_mask(svt, rgt, 60.0, 200.0, 15, -6.0)
    pad = torch.nn.functional.pad
    svp = pad(svt, (0, 1), value=float("nan")).contiguous()
    rgp = pad(rgt, (0, 1), value=1.0e6).contiguous()   # (a NaN would win both argmins: the layer would be empty)
    ref = ops.attenuated_mask(svp, rgp, 60.0, 200.0, 15, -6.0)[:, :, :S]
    torch.cuda.synchronize()
    a, b = new.cpu().numpy()[:, :, 0], ref.cpu().numpy()[:, :, 0]
    print(dt, "pings flagged", int(a.sum()), int(b.sum()), "differ", int((a != b).sum()), np.argwhere(a != b)[:8].tolist())

# a depth with heave in it: the layer limits move by a sample or two from ping to ping (shifted inside the ring)
depth2 = np.broadcast_to(1.0 + 0.5 * np.arange(S), (C, P, S)) + 3.0 * np.sin(np.arange(P) / 7.0)[None, :, None]
svt, rgt = torch.from_numpy(sv).cuda(), torch.from_numpy(depth2.copy()).cuda()
new = ops.attenuated_mask(svt, rgt, 60.0, 200.0, 15, -6.0)
ref = ops.attenuated_mask(torch.nn.functional.pad(svt, (0, 1), value=float("nan")).contiguous(),
                          torch.nn.functional.pad(rgt, (0, 1), value=1.0e6).contiguous(), 60.0, 200.0, 15, -6.0)[:, :, :S]
a, b = new.cpu().numpy()[:, :, 0], ref.cpu().numpy()[:, :, 0]
print("heave: pings flagged", int(a.sum()), int(b.sum()), "differ", int((a != b).sum()))
t = ops.Timer()
t.start(); ops.attenuated_mask(svt, rgt, 60.0, 200.0, 15, -6.0); t.stop()
print("heave ms", t.elapsed_ms())
