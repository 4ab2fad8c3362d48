"""PyTorch-CPU fp32 restatement of the reference's library calls.  TEST INFRASTRUCTURE ONLY.

The reference (server/model.py:18-19,47,59) runs facenet_pytorch==2.6.0's ``MTCNN`` and
``InceptionResnetV1`` on torch CPU.  That package is absent here, so this
file restates its published module structure (RECALLED) on plain torch ops
with the SAME ``state_dict`` key layout, for two purposes only:

* cross-check the C oracle (oracle/trl_oracle.c) and the weight packer against torch's own
  ``conv2d`` / ``max_pool2d(ceil_mode)`` / ``adaptive_avg_pool2d`` / ``batch_norm`` / ``normalize``
  at fp32 tolerance (the C oracle fixes an accumulation order, torch does not), and
* serve as bench.py's ``cpu_baseline`` ("restated reference CPU path": torch CPU, one frame at a
  time, exactly how model.py drives the library).

PARITY UNPINNED: nothing here has been compared with the real package.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _sd(d):
    return {k: torch.as_tensor(np.asarray(v)) for k, v in d.items()}


class PNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 10, 3); self.prelu1 = nn.PReLU(10)
        self.pool1 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.conv2 = nn.Conv2d(10, 16, 3); self.prelu2 = nn.PReLU(16)
        self.conv3 = nn.Conv2d(16, 32, 3); self.prelu3 = nn.PReLU(32)
        self.conv4_1 = nn.Conv2d(32, 2, 1); self.conv4_2 = nn.Conv2d(32, 4, 1)

    def forward(self, x):
        x = self.pool1(self.prelu1(self.conv1(x)))
        x = self.prelu2(self.conv2(x))
        x = self.prelu3(self.conv3(x))
        return self.conv4_2(x), F.softmax(self.conv4_1(x), dim=1)


class RNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 28, 3); self.prelu1 = nn.PReLU(28); self.pool1 = nn.MaxPool2d(3, 2, ceil_mode=True)
        self.conv2 = nn.Conv2d(28, 48, 3); self.prelu2 = nn.PReLU(48); self.pool2 = nn.MaxPool2d(3, 2, ceil_mode=True)
        self.conv3 = nn.Conv2d(48, 64, 2); self.prelu3 = nn.PReLU(64)
        self.dense4 = nn.Linear(576, 128); self.prelu4 = nn.PReLU(128)
        self.dense5_1 = nn.Linear(128, 2); self.dense5_2 = nn.Linear(128, 4)

    def forward(self, x):
        x = self.pool1(self.prelu1(self.conv1(x)))
        x = self.pool2(self.prelu2(self.conv2(x)))
        x = self.prelu3(self.conv3(x))
        x = x.permute(0, 3, 2, 1).contiguous()
        x = self.prelu4(self.dense4(x.view(x.shape[0], -1)))
        return self.dense5_2(x), F.softmax(self.dense5_1(x), dim=1)


class ONet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 32, 3); self.prelu1 = nn.PReLU(32); self.pool1 = nn.MaxPool2d(3, 2, ceil_mode=True)
        self.conv2 = nn.Conv2d(32, 64, 3); self.prelu2 = nn.PReLU(64); self.pool2 = nn.MaxPool2d(3, 2, ceil_mode=True)
        self.conv3 = nn.Conv2d(64, 64, 3); self.prelu3 = nn.PReLU(64); self.pool3 = nn.MaxPool2d(2, 2, ceil_mode=True)
        self.conv4 = nn.Conv2d(64, 128, 2); self.prelu4 = nn.PReLU(128)
        self.dense5 = nn.Linear(1152, 256); self.prelu5 = nn.PReLU(256)
        self.dense6_1 = nn.Linear(256, 2); self.dense6_2 = nn.Linear(256, 4); self.dense6_3 = nn.Linear(256, 10)

    def forward(self, x):
        x = self.pool1(self.prelu1(self.conv1(x)))
        x = self.pool2(self.prelu2(self.conv2(x)))
        x = self.pool3(self.prelu3(self.conv3(x)))
        x = self.prelu4(self.conv4(x))
        x = x.permute(0, 3, 2, 1).contiguous()
        x = self.prelu5(self.dense5(x.view(x.shape[0], -1)))
        return self.dense6_2(x), self.dense6_3(x), F.softmax(self.dense6_1(x), dim=1)


class BasicConv2d(nn.Module):
    def __init__(self, i, o, k, s=1, p=0):
        super().__init__()
        self.conv = nn.Conv2d(i, o, k, s, p, bias=False)
        self.bn = nn.BatchNorm2d(o, eps=0.001, momentum=0.1, affine=True)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class Block35(nn.Module):
    def __init__(self, scale):
        super().__init__()
        self.scale = scale
        self.branch0 = BasicConv2d(256, 32, 1)
        self.branch1 = nn.Sequential(BasicConv2d(256, 32, 1), BasicConv2d(32, 32, 3, 1, 1))
        self.branch2 = nn.Sequential(BasicConv2d(256, 32, 1), BasicConv2d(32, 32, 3, 1, 1), BasicConv2d(32, 32, 3, 1, 1))
        self.conv2d = nn.Conv2d(96, 256, 1)

    def forward(self, x):
        out = self.conv2d(torch.cat((self.branch0(x), self.branch1(x), self.branch2(x)), 1))
        return F.relu(out * self.scale + x)


class Block17(nn.Module):
    def __init__(self, scale):
        super().__init__()
        self.scale = scale
        self.branch0 = BasicConv2d(896, 128, 1)
        self.branch1 = nn.Sequential(BasicConv2d(896, 128, 1), BasicConv2d(128, 128, (1, 7), 1, (0, 3)),
                                     BasicConv2d(128, 128, (7, 1), 1, (3, 0)))
        self.conv2d = nn.Conv2d(256, 896, 1)

    def forward(self, x):
        out = self.conv2d(torch.cat((self.branch0(x), self.branch1(x)), 1))
        return F.relu(out * self.scale + x)


class Block8(nn.Module):
    def __init__(self, scale=1.0, noReLU=False):
        super().__init__()
        self.scale, self.noReLU = scale, noReLU
        self.branch0 = BasicConv2d(1792, 192, 1)
        self.branch1 = nn.Sequential(BasicConv2d(1792, 192, 1), BasicConv2d(192, 192, (1, 3), 1, (0, 1)),
                                     BasicConv2d(192, 192, (3, 1), 1, (1, 0)))
        self.conv2d = nn.Conv2d(384, 1792, 1)

    def forward(self, x):
        out = self.conv2d(torch.cat((self.branch0(x), self.branch1(x)), 1))
        out = out * self.scale + x
        return out if self.noReLU else F.relu(out)


class Mixed_6a(nn.Module):
    def __init__(self):
        super().__init__()
        self.branch0 = BasicConv2d(256, 384, 3, 2)
        self.branch1 = nn.Sequential(BasicConv2d(256, 192, 1), BasicConv2d(192, 192, 3, 1, 1), BasicConv2d(192, 256, 3, 2))
        self.branch2 = nn.MaxPool2d(3, 2)

    def forward(self, x):
        return torch.cat((self.branch0(x), self.branch1(x), self.branch2(x)), 1)


class Mixed_7a(nn.Module):
    def __init__(self):
        super().__init__()
        self.branch0 = nn.Sequential(BasicConv2d(896, 256, 1), BasicConv2d(256, 384, 3, 2))
        self.branch1 = nn.Sequential(BasicConv2d(896, 256, 1), BasicConv2d(256, 256, 3, 2))
        self.branch2 = nn.Sequential(BasicConv2d(896, 256, 1), BasicConv2d(256, 256, 3, 1, 1), BasicConv2d(256, 256, 3, 2))
        self.branch3 = nn.MaxPool2d(3, 2)

    def forward(self, x):
        return torch.cat((self.branch0(x), self.branch1(x), self.branch2(x), self.branch3(x)), 1)


class InceptionResnetV1(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv2d_1a = BasicConv2d(3, 32, 3, 2); self.conv2d_2a = BasicConv2d(32, 32, 3)
        self.conv2d_2b = BasicConv2d(32, 64, 3, 1, 1); self.maxpool_3a = nn.MaxPool2d(3, 2)
        self.conv2d_3b = BasicConv2d(64, 80, 1); self.conv2d_4a = BasicConv2d(80, 192, 3)
        self.conv2d_4b = BasicConv2d(192, 256, 3, 2)
        self.repeat_1 = nn.Sequential(*[Block35(0.17) for _ in range(5)])
        self.mixed_6a = Mixed_6a()
        self.repeat_2 = nn.Sequential(*[Block17(0.10) for _ in range(10)])
        self.mixed_7a = Mixed_7a()
        self.repeat_3 = nn.Sequential(*[Block8(0.20) for _ in range(5)])
        self.block8 = Block8(noReLU=True)
        self.avgpool_1a = nn.AdaptiveAvgPool2d(1)
        self.last_linear = nn.Linear(1792, 512, bias=False)
        self.last_bn = nn.BatchNorm1d(512, eps=0.001, momentum=0.1, affine=True)

    def forward(self, x):
        for m in (self.conv2d_1a, self.conv2d_2a, self.conv2d_2b, self.maxpool_3a, self.conv2d_3b, self.conv2d_4a,
                  self.conv2d_4b, self.repeat_1, self.mixed_6a, self.repeat_2, self.mixed_7a, self.repeat_3,
                  self.block8, self.avgpool_1a):
            x = m(x)
        x = self.last_bn(self.last_linear(x.view(x.shape[0], -1)))
        return F.normalize(x, p=2, dim=1)


# ---- detect_face restated on torch / numpy ops (RECALLED: facenet_pytorch utils/detect_face.py) ----

def _margin(audit, key, values, thr):
    """Near-threshold audit: how close did any compared value come to its decision threshold?"""
    if audit is None or len(values) == 0:
        return
    d = np.abs(np.asarray(values, np.float64) - float(np.float32(thr)))
    a = audit.setdefault(key, {"n": 0, "min_margin": float("inf"), "within_1e-5": 0, "within_1e-6": 0})
    a["n"] += int(d.size); a["min_margin"] = min(a["min_margin"], float(d.min()))
    a["within_1e-5"] += int((d <= 1e-5).sum()); a["within_1e-6"] += int((d <= 1e-6).sum())


def _nms_iou(boxes: torch.Tensor, scores: torch.Tensor, thr: float, audit=None, key="iou") -> torch.Tensor:
    """torchvision.ops.nms semantics (stable descending sort, greedy, strict >)."""
    b = boxes.numpy(); s = scores.numpy()
    order = np.argsort(-s, kind="stable")
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = np.zeros(len(s), bool); keep = []
    for _i, i in enumerate(order):
        if sup[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(b[i, 0], b[rest, 0]); yy1 = np.maximum(b[i, 1], b[rest, 1])
        xx2 = np.minimum(b[i, 2], b[rest, 2]); yy2 = np.minimum(b[i, 3], b[rest, 3])
        inter = np.maximum(np.float32(0), xx2 - xx1) * np.maximum(np.float32(0), yy2 - yy1)
        ovr = inter / (area[i] + area[rest] - inter)
        _margin(audit, key, ovr[~sup[rest]], thr)
        sup[rest[ovr > np.float32(thr)]] = True
    return torch.as_tensor(np.array(keep, dtype=np.int64))


def _nms_min(boxes: np.ndarray, scores: np.ndarray, thr: float, audit=None, key="iom") -> np.ndarray:
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    I = np.argsort(scores, kind="stable")
    pick = []
    while I.size > 0:
        i = I[-1]; pick.append(i); idx = I[:-1]
        w = np.maximum(np.float32(0), np.minimum(x2[i], x2[idx]) - np.maximum(x1[i], x1[idx]) + 1)
        h = np.maximum(np.float32(0), np.minimum(y2[i], y2[idx]) - np.maximum(y1[i], y1[idx]) + 1)
        o = (w * h) / np.minimum(area[i], area[idx])
        _margin(audit, key, o, thr)
        I = idx[o <= np.float32(thr)]
    return np.array(pick, dtype=np.int64)


def _rerec(b):
    h = b[:, 3] - b[:, 1]; w = b[:, 2] - b[:, 0]; l = torch.max(w, h)
    b[:, 0] = b[:, 0] + w * 0.5 - l * 0.5
    b[:, 1] = b[:, 1] + h * 0.5 - l * 0.5
    b[:, 2:4] = b[:, :2] + l.repeat(2, 1).permute(1, 0)
    return b


def _bbreg(b, reg):
    w = b[:, 2] - b[:, 0] + 1; h = b[:, 3] - b[:, 1] + 1
    b[:, :4] = torch.stack([b[:, 0] + reg[:, 0] * w, b[:, 1] + reg[:, 1] * h,
                            b[:, 2] + reg[:, 2] * w, b[:, 3] + reg[:, 3] * h]).permute(1, 0)
    return b


def _pad(boxes, w, h):
    b = boxes.trunc().int().numpy()
    x, y, ex, ey = b[:, 0].copy(), b[:, 1].copy(), b[:, 2].copy(), b[:, 3].copy()
    x[x < 1] = 1; y[y < 1] = 1; ex[ex > w] = w; ey[ey > h] = h
    return y, ey, x, ex


class TorchRef:
    """MTCNN().detect + InceptionResnetV1().eval() exactly as model.py:18-19,47,59 drive them."""

    def __init__(self, pnet_sd, rnet_sd, onet_sd, facenet_sd, threads: int | None = None, min_face_size: int = 20):
        if threads:
            torch.set_num_threads(int(threads))
        self.pnet, self.rnet, self.onet, self.facenet = PNet(), RNet(), ONet(), InceptionResnetV1()
        self.pnet.load_state_dict(_sd(pnet_sd)); self.rnet.load_state_dict(_sd(rnet_sd))
        self.onet.load_state_dict(_sd(onet_sd)); self.facenet.load_state_dict(_sd(facenet_sd), strict=False)
        for m in (self.pnet, self.rnet, self.onet, self.facenet):
            m.eval()
        self.minsize, self.thr, self.factor = int(min_face_size), (0.6, 0.7, 0.7), 0.709

    @torch.no_grad()
    def detect(self, frame: np.ndarray, trace: dict | None = None):
        imgs = torch.as_tensor(frame.copy()).unsqueeze(0).permute(0, 3, 1, 2).float()
        h, w = imgs.shape[2:4]
        m = 12.0 / self.minsize; minl = min(h, w) * m; scale_i = m; scales = []
        while minl >= 12:
            scales.append(scale_i); scale_i *= self.factor; minl *= self.factor
        allb = []
        audit = trace.setdefault("audit", {}) if trace is not None else None
        if trace is not None:
            trace["cand_cells"], trace["keep_cells"] = [], []
        for scale in scales:
            im = F.interpolate(imgs, size=(int(h * scale + 1), int(w * scale + 1)), mode="area")
            im = (im - 127.5) * 0.0078125
            reg, probs = self.pnet(im)
            p = probs[0, 1]
            mask = p >= self.thr[0]
            idx = mask.nonzero()
            bb = idx.float().flip(1)
            q1 = ((2 * bb + 1) / scale).floor(); q2 = ((2 * bb + 12) / scale).floor()
            r = reg[0].permute(1, 2, 0)[mask]
            boxes = torch.cat([q1, q2, p[mask].unsqueeze(1), r], 1)
            cells = (idx[:, 0] * p.shape[1] + idx[:, 1]).numpy()
            _margin(audit, "pnet_prob_vs_thr0", p.numpy().reshape(-1), self.thr[0])
            pick = _nms_iou(boxes[:, :4], boxes[:, 4], 0.5, audit, "stage1a_iou_vs_0.5") if len(boxes) else torch.zeros(0, dtype=torch.int64)
            if trace is not None:
                trace["cand_cells"].append(cells); trace["keep_cells"].append(cells[pick.numpy()])
            if len(boxes):
                allb.append(boxes[pick])
        if not allb:
            return None, None
        boxes = torch.cat(allb, 0)
        pick = _nms_iou(boxes[:, :4], boxes[:, 4], 0.7, audit, "stage1b_iou_vs_0.7")
        boxes = boxes[pick]
        if trace is not None:
            trace["pick1"] = pick.numpy().copy()          # positions in the concatenation of the per-scale keeps
        regw = boxes[:, 2] - boxes[:, 0]; regh = boxes[:, 3] - boxes[:, 1]
        boxes = torch.stack([boxes[:, 0] + boxes[:, 5] * regw, boxes[:, 1] + boxes[:, 6] * regh,
                             boxes[:, 2] + boxes[:, 7] * regw, boxes[:, 3] + boxes[:, 8] * regh, boxes[:, 4]]).permute(1, 0)
        boxes = _rerec(boxes.contiguous())
        if trace is not None:
            trace["boxes1"] = boxes.numpy().copy()

        def crops(boxes, size):
            y, ey, x, ex = _pad(boxes, w, h)
            out = []
            for k in range(len(y)):
                if ey[k] > y[k] - 1 and ex[k] > x[k] - 1:
                    out.append(F.interpolate(imgs[:, :, y[k] - 1:ey[k], x[k] - 1:ex[k]], size=(size, size), mode="area"))
            return (torch.cat(out, 0) - 127.5) * 0.0078125

        reg, prob = self.rnet(crops(boxes, 24))
        score = prob[:, 1]; ipass = score > self.thr[1]
        _margin(audit, "rnet_prob_vs_thr1", score.numpy(), self.thr[1])
        if trace is not None:
            trace["pass2"] = ipass.nonzero()[:, 0].numpy().copy()
        boxes = torch.cat((boxes[ipass, :4], score[ipass].unsqueeze(1)), 1); mv = reg[ipass]
        if len(boxes) == 0:
            return None, None
        pick = _nms_iou(boxes[:, :4], boxes[:, 4], 0.7, audit, "stage2_iou_vs_0.7")
        if trace is not None:
            trace["pick2"] = trace["pass2"][pick.numpy()]     # indices into boxes1
        boxes = _rerec(_bbreg(boxes[pick], mv[pick]))
        if trace is not None:
            trace["boxes2"] = boxes.numpy().copy()
        reg, pts, prob = self.onet(crops(boxes, 48))
        score = prob[:, 1]; ipass = score > self.thr[2]
        _margin(audit, "onet_prob_vs_thr2", score.numpy(), self.thr[2])
        if trace is not None:
            trace["pass3"] = ipass.nonzero()[:, 0].numpy().copy()
        boxes = torch.cat((boxes[ipass, :4], score[ipass].unsqueeze(1)), 1); mv = reg[ipass]
        if len(boxes) == 0:
            return None, None
        boxes = _bbreg(boxes, mv).numpy()
        pick3 = _nms_min(boxes[:, :4], boxes[:, 4], 0.7, audit, "stage3_iom_vs_0.7")
        if trace is not None:
            trace["pick3"] = trace["pass3"][pick3]            # indices into boxes2
        boxes = boxes[pick3]
        if trace is not None:
            trace["boxes3"] = boxes.copy()
        order = np.argsort((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]), kind="stable")[::-1]
        boxes = boxes[order]
        return boxes[:, :4], boxes[:, 4]

    @torch.no_grad()
    def embed(self, face_u8_hwc: np.ndarray) -> np.ndarray:
        """to_tensor (u8 HWC -> f32 CHW / 255) + facenet, model.py:58-59."""
        x = torch.as_tensor(face_u8_hwc.copy()).permute(2, 0, 1).float().div(255).unsqueeze(0)
        return self.facenet(x).numpy().reshape(-1)
