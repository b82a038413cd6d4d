"""Synthetic Duckietown-like frames + 4-class masks, generated with a fixed seed.

Frame: 480x640 BGR uint8-valued image = sky / ground split at a random horizon, grey road trapezoid, white
and yellow lane stripes, +-8 uniform noise; area-averaged to HxW (default 120x160) and normalised exactly as
the reference input pipeline does: (px/255 - mean)/std with ImageNet mean/std applied in stored BGR order
(dataManagement/myTransforms.py:18, myDatasets.py:51).  Label: {0 background, 1 right lane, 2 left lane,
3 obstacle} from the same geometry (utils/createRealDB.py:12-17); every 8th sample has no obstacle.
Input plumbing only (torch ops, outside any timed region); not part of the hot path.
"""
import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def make_batch(n, h=120, w=160, seed=42, first_index=0, device="cpu"):
    """Returns (x float32 [n,3,h,w], y int64 [n,h,w]); sample i depends only on seed + first_index + i."""
    xs, ys = [], []
    fh, fw = 4 * h, 4 * w
    yy = torch.arange(fh, dtype=torch.float32).view(fh, 1)
    xx = torch.arange(fw, dtype=torch.float32).view(1, fw)
    for i in range(n):
        g = torch.Generator().manual_seed(seed + first_index + i)
        r = torch.rand(8, generator=g)
        horizon = (80 + 120 * float(r[0])) * fh / 480.0
        cx = fw * (0.4 + 0.2 * float(r[1]))
        t = ((yy - horizon) / (fh - horizon)).clamp(0, 1)          # 0 at horizon, 1 at bottom
        half = (0.04 + 0.46 * t) * fw                               # road half-width grows toward the camera
        below = yy > horizon
        on_road = below & ((xx - cx).abs() < half)
        img = torch.empty(3, fh, fw)
        sky = torch.tensor([200.0, 170.0, 120.0]).view(3, 1, 1) * (0.8 + 0.2 * float(r[2]))
        grass = torch.tensor([60.0, 130.0, 70.0]).view(3, 1, 1) * (0.8 + 0.2 * float(r[3]))
        img[:] = torch.where(below.expand(3, fh, fw), grass.expand(3, fh, fw), sky.expand(3, fh, fw))
        img = torch.where(on_road.expand(3, fh, fw), torch.full_like(img, 90.0), img)
        lab = torch.zeros(fh, fw, dtype=torch.int64)
        right = on_road & (xx > cx)
        left = on_road & (xx <= cx)
        lab[right] = 1
        lab[left] = 2
        stripe_w = (0.006 + 0.02 * t) * fw
        edge = below & (((xx - cx).abs() - half).abs() < stripe_w)           # white side lines
        centre = below & ((xx - cx).abs() < stripe_w) & (((yy / (fh / 12.0)).floor() % 2) == 0)  # dashed yellow
        img = torch.where(edge.expand(3, fh, fw), torch.full_like(img, 235.0), img)
        yellow = torch.tensor([40.0, 210.0, 230.0]).view(3, 1, 1).expand(3, fh, fw)
        img = torch.where(centre.expand(3, fh, fw), yellow, img)
        if (first_index + i) % 8 != 7:  # obstacle (a box on the road), absent in every 8th sample
            oy = horizon + (fh - horizon) * (0.3 + 0.5 * float(r[4]))
            ox = cx + (float(r[5]) - 0.5) * fw * 0.3
            osz = fh * (0.05 + 0.08 * float(r[6]))
            box = ((yy - oy).abs() < osz) & ((xx - ox).abs() < osz * 0.8)
            img = torch.where(box.expand(3, fh, fw), torch.tensor([30.0, 40.0, 200.0]).view(3, 1, 1).expand(3, fh, fw),
                              img)
            lab[box] = 3
        img = (img + (torch.rand(3, fh, fw, generator=g) * 16 - 8)).clamp(0, 255).round()
        small = torch.nn.functional.avg_pool2d(img.unsqueeze(0), 4).squeeze(0)   # area average of 4x4 blocks
        lab_small = lab[2::4, 2::4].contiguous()                                 # nearest for labels
        mean = torch.tensor(MEAN).view(3, 1, 1)
        std = torch.tensor(STD).view(3, 1, 1)
        xs.append((small / 255.0 - mean) / std)
        ys.append(lab_small)
    x = torch.stack(xs).float().contiguous()
    y = torch.stack(ys).contiguous()
    return x.to(device), y.to(device)
