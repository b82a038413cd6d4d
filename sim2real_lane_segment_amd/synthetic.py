"""Synthetic Duckietown-like frames + 4-class masks, generated with a fixed seed.

Frame: 480x640 BGR uint8-valued image = sky / ground split at a random horizon, grey road trapezoid, white
and yellow lane stripes, +-8 uniform noise; area-averaged to HxW (default 120x160) and normalised exactly as
the reference input pipeline does: (px/255 - mean)/std with ImageNet mean/std applied in stored BGR order
(dataManagement/myTransforms.py:18, myDatasets.py:51).  Label: {0 background, 1 right lane, 2 left lane,
3 obstacle} from the same geometry (utils/createRealDB.py:12-17); every 8th sample has no obstacle.
Two domains (BASELINE.json configs[4], dataManagement/dataModules.py:64-85): domain 0 = the simulator look above;
domain 1 = a "real camera" look of the same geometry (indoor grey background, darker textured road, dimmer paint,
vignetting, three times the sensor noise).  make_two_domain_batch draws every sample's domain the way the reference's
TwoDomainDM.train_dataloader does: a WeightedRandomSampler over the concatenated source + target sets with weights
1/len(source) and 1/len(target), i.e. each domain with probability 1/2 whatever the set sizes.
Input plumbing only (torch ops, outside any timed region); not part of the hot path.
"""
import torch

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def make_batch(n, h=120, w=160, seed=42, first_index=0, device="cpu", domains=None, frames_u8=False, work_device="cpu"):
    """Returns (x float32 [n,3,h,w], y int64 [n,h,w]); sample i depends only on seed + first_index + i (and on
    domains[i] in {0, 1} when given).  frames_u8=True returns the full-size frames instead, as the reference's datasets
    hand them to MyTransform (myDatasets.py:45-61): (uint8 [n,4h,4w,3] in stored BGR order, uint8 labels [n,4h,4w]).
    work_device: where the per-pixel drawing runs (the random numbers always come from the CPU generator, so a sample
    is the same wherever it is drawn, up to the last-bit rounding of the vignette)."""
    xs, ys = [], []
    fh, fw = 4 * h, 4 * w
    wd = torch.device(work_device)
    yy = torch.arange(fh, dtype=torch.float32, device=wd).view(fh, 1)
    xx = torch.arange(fw, dtype=torch.float32, device=wd).view(1, fw)

    def T(vals):
        return torch.tensor(vals, device=wd).view(3, 1, 1)
    for i in range(n):
        g = torch.Generator().manual_seed(seed + first_index + i)
        r = torch.rand(8, generator=g)
        horizon = (80 + 120 * float(r[0])) * fh / 480.0
        cx = fw * (0.4 + 0.2 * float(r[1]))
        t = ((yy - horizon) / (fh - horizon)).clamp(0, 1)          # 0 at horizon, 1 at bottom
        half = (0.04 + 0.46 * t) * fw                               # road half-width grows toward the camera
        below = yy > horizon
        on_road = below & ((xx - cx).abs() < half)
        img = torch.empty(3, fh, fw, device=wd)
        real = domains is not None and int(domains[i]) == 1
        if real:  # indoor wall / floor instead of sky / grass, darker road with a coarse texture
            sky = T([150.0, 150.0, 155.0]) * (0.7 + 0.3 * float(r[2]))
            grass = T([105.0, 110.0, 120.0]) * (0.7 + 0.3 * float(r[3]))
            tex = torch.nn.functional.interpolate(torch.rand(1, 1, fh // 8, fw // 8, generator=g).to(wd), size=(fh, fw),
                                                  mode="nearest")[0] * 30.0 - 15.0
            road = (55.0 + tex).expand(3, fh, fw)
        else:
            sky = T([200.0, 170.0, 120.0]) * (0.8 + 0.2 * float(r[2]))
            grass = T([60.0, 130.0, 70.0]) * (0.8 + 0.2 * float(r[3]))
            road = torch.full((3, fh, fw), 90.0, device=wd)
        img[:] = torch.where(below.expand(3, fh, fw), grass.expand(3, fh, fw), sky.expand(3, fh, fw))
        img = torch.where(on_road.expand(3, fh, fw), road, img)
        lab = torch.zeros(fh, fw, dtype=torch.int64, device=wd)
        right = on_road & (xx > cx)
        left = on_road & (xx <= cx)
        lab[right] = 1
        lab[left] = 2
        stripe_w = (0.006 + 0.02 * t) * fw
        edge = below & (((xx - cx).abs() - half).abs() < stripe_w)           # white side lines
        centre = below & ((xx - cx).abs() < stripe_w) & (((yy / (fh / 12.0)).floor() % 2) == 0)  # dashed yellow
        img = torch.where(edge.expand(3, fh, fw), torch.full_like(img, 190.0 if real else 235.0), img)
        yellow = (T([60.0, 170.0, 185.0]) if real else T([40.0, 210.0, 230.0])).expand(3, fh, fw)
        img = torch.where(centre.expand(3, fh, fw), yellow, img)
        if (first_index + i) % 8 != 7:  # obstacle (a box on the road), absent in every 8th sample
            oy = horizon + (fh - horizon) * (0.3 + 0.5 * float(r[4]))
            ox = cx + (float(r[5]) - 0.5) * fw * 0.3
            osz = fh * (0.05 + 0.08 * float(r[6]))
            box = ((yy - oy).abs() < osz) & ((xx - ox).abs() < osz * 0.8)
            img = torch.where(box.expand(3, fh, fw), T([30.0, 40.0, 200.0]).expand(3, fh, fw), img)
            lab[box] = 3
        if real:  # vignetting + 3x the sensor noise
            vig = 1.0 - 0.35 * (((yy - fh / 2) / (fh / 2)) ** 2 + ((xx - fw / 2) / (fw / 2)) ** 2) / 2
            img = img * vig
            img = (img + (torch.rand(3, fh, fw, generator=g).to(wd) * 48 - 24)).clamp(0, 255).round()
        else:
            img = (img + (torch.rand(3, fh, fw, generator=g).to(wd) * 16 - 8)).clamp(0, 255).round()
        if frames_u8:
            xs.append(img.permute(1, 2, 0).to(torch.uint8))
            ys.append(lab.to(torch.uint8))
            continue
        small = torch.nn.functional.avg_pool2d(img.unsqueeze(0), 4).squeeze(0)   # area average of 4x4 blocks
        lab_small = lab[2::4, 2::4].contiguous()                                 # nearest for labels
        mean = T(list(MEAN))
        std = T(list(STD))
        xs.append((small / 255.0 - mean) / std)
        ys.append(lab_small)
    if frames_u8:
        return torch.stack(xs).contiguous().to(device), torch.stack(ys).contiguous().to(device)
    x = torch.stack(xs).float().contiguous()
    y = torch.stack(ys).contiguous()
    return x.to(device), y.to(device)


def two_domain_indices(source_len, target_len, num_samples, seed):
    """Indices into ConcatDataset([source, target]) drawn as TwoDomainDM.train_dataloader draws them
    (dataManagement/dataModules.py:79-85): torch.multinomial over weights 1/len(source) | 1/len(target) with replacement
    (what torch.utils.data.WeightedRandomSampler does).  Returns (index int64 [num_samples], domain int64 [num_samples])."""
    w = torch.cat([torch.full((source_len,), 1.0 / source_len, dtype=torch.double),
                   torch.full((target_len,), 1.0 / target_len, dtype=torch.double)])
    g = torch.Generator().manual_seed(seed)
    idx = torch.multinomial(w, num_samples, True, generator=g)
    return idx, (idx >= source_len).long()


def make_two_domain_batch(n, h=120, w=160, seed=42, source_len=1000, target_len=100, device="cpu"):
    """One training batch of the combined sim + real configuration: sample i is frame idx[i] of its domain's set, the
    domains mixed 50/50 by the weighted sampler above.  Returns (x, y, domain)."""
    idx, dom = two_domain_indices(source_len, target_len, n, seed)
    xs, ys = [], []
    for i in range(n):
        local = int(idx[i]) - (source_len if int(dom[i]) else 0)
        x, y = make_batch(1, h, w, seed=seed + 7919 * int(dom[i]), first_index=local, domains=[int(dom[i])])
        xs.append(x)
        ys.append(y)
    return torch.cat(xs).to(device), torch.cat(ys).to(device), dom
