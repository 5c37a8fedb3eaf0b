"""Independent cross-check of the C oracle against torch-CPU conv2d.

TEST INFRASTRUCTURE ONLY.  torch here is library code, not reference code: the
same graph (SuperResolution.cpp:172-275) is evaluated with float64 convolutions
and an fp16 rounding after every layer, and compared with oracle/sr_oracle.c on
the bundled images.  The two differ only where a float64 sum and the oracle's
fp32 running sum fall on different sides of an fp16 rounding boundary.

Also reports what the two folded forms of conv_up1 cost against the direct form
(up1_mode=0): the four-phase 3x3 form of the layer-wise kernels (up1_mode=1) and the
row-shared form of the fused kernel (up1_mode=2), which is the evidence behind the
tolerance the HIP path is tested to.

Run:  python oracle/crosscheck_torch.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sro  # noqa: E402

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                      "superresolution_amd", "Assets")


def f16(t):
    return t.to(torch.float16).to(torch.float64)


def torch_graph(W, rgba):
    x = torch.from_numpy(rgba[..., :3].astype(np.float32) / np.float32(255.0))
    x = f16(x.permute(2, 0, 1)[None])
    y = F.interpolate(x, scale_factor=2, mode="nearest")
    outs = {"input": x}
    cur = x
    for li, (conv, scale, shift, sizes) in enumerate(sro.LAYERS):
        f, b = W.create_weight_tensors(conv, scale, shift, sizes)
        wt = torch.from_numpy(f.view(np.float16).astype(np.float64)).reshape(sizes)
        bt = torch.from_numpy(b.view(np.float16).astype(np.float64)) if b is not None else None
        if li == 3:
            cur = F.interpolate(cur, scale_factor=2, mode="nearest")
        cur = F.conv2d(cur, wt, bt, padding=sizes[2] // 2)
        if b is not None:
            cur = torch.relu(cur)
        cur = f16(cur)
        outs[sro.TENSOR_NAMES[li + 1]] = cur
    out = f16(cur + y)
    outs["output"] = out
    img = torch.floor(out.clamp(0, 1).to(torch.float32) * 255.0 + 0.5).to(torch.uint8)
    return outs, img[0].permute(1, 2, 0).numpy()


def compare_images(a, b):
    d = np.abs(a[..., :3].astype(np.int32) - b[..., :3].astype(np.int32))
    return dict(equal=float((d == 0).mean()), le1=float((d <= 1).mean()), max=int(d.max()))


def main():
    torch.set_num_threads(8)
    W = sro.Weights(os.path.join(ASSETS, "weights.bin"))
    direct = sro.Model(W, 0, 0)
    phases = sro.Model(W, 0, 1)
    rowshared = sro.Model(W, 0, 2)
    for name in ["Sashimi128.png", "xiaoyao.png", "Sashimi.png"]:
        rgba = sro.load_rgba(os.path.join(ASSETS, name))
        o_img = direct.run(rgba, keep_tensors=True)
        t_outs, t_img = torch_graph(W, rgba)
        print(f"== {name} {rgba.shape[1]}x{rgba.shape[0]}")
        for tid, tname in enumerate(sro.TENSOR_NAMES):
            o = direct.tensor(tid)
            t = t_outs[tname][0].numpy()
            d = np.abs(o.astype(np.float64) - t)
            print(f"   oracle vs torch64  {tname:9s} max_abs {d.max():.3e}  differing {float((d > 0).mean()):.2e}")
        print("   oracle vs torch64  image", compare_images(o_img, t_img))
        p_img = phases.run(rgba, keep_tensors=True)
        for tid in (sro.T_CONV_UP1, sro.T_CONV4, sro.T_CONV5, sro.T_CONV6, sro.T_OUTPUT):
            d = np.abs(direct.tensor(tid) - phases.tensor(tid))
            print(f"   phases vs direct   {sro.TENSOR_NAMES[tid]:9s} max_abs {d.max():.3e}  mean_abs {d.mean():.3e}")
        print("   phases vs direct   image", compare_images(o_img, p_img))
        r_img = rowshared.run(rgba, keep_tensors=True)
        for tid in (sro.T_CONV_UP1, sro.T_CONV4, sro.T_CONV5, sro.T_CONV6, sro.T_OUTPUT):
            d = np.abs(direct.tensor(tid) - rowshared.tensor(tid))
            print(f"   row-shared vs direct {sro.TENSOR_NAMES[tid]:9s} max_abs {d.max():.3e}  mean_abs {d.mean():.3e}")
        print("   row-shared vs direct image", compare_images(o_img, r_img))


if __name__ == "__main__":
    main()
