"""CPU stand-in for HipShardEngine built from the oracle: TEST INFRASTRUCTURE ONLY.  It lets the exchange logic of
cameraShard.CameraShardedBev (boxes, group order, saturation, the balance round) run under gloo without a GPU, and it is
the checker for the GPU runs of the same class.  Follows surroundBEV.py:57-79 and :312-325 through oracle/oracle.py."""
import numpy as np

from oracle import oracle as O

CAMS = ("front", "back", "left", "right")


def module_cfg():
    from cameracalibration_amd.SurroundBirdEyeView import surroundBEV as SB

    a = SB.BevGenerator.get_args()
    return {k: getattr(a, k) for k in ("FRAME_WIDTH", "FRAME_HEIGHT", "BEV_WIDTH", "BEV_HEIGHT", "CAR_WIDTH", "CAR_HEIGHT",
                                        "FOCAL_SCALE", "SIZE_SCALE")}


def mask_box(masks, bw, bh):
    ys, xs = np.nonzero(np.maximum.reduce(masks))
    if ys.size == 0:
        return (0, 0, min(bw, 4), 1)
    x0, x1 = int(xs.min()) & ~3, min(bw, (int(xs.max()) + 1 + 3) & ~3)
    return (x0, int(ys.min()), x1, int(ys.max()) + 1)


class OracleShardEngine:
    def __init__(self, rig_list, cams, blend, balance, device=0):
        O.build()
        self.cfg = c = module_cfg()
        self.cams, self.blend, self.balance = tuple(cams), bool(blend), bool(balance)
        self.cameras = {k: O.RefCamera(*rig_list[k], c) for k in self.cams}
        geo = (c["BEV_WIDTH"], c["BEV_HEIGHT"], c["CAR_WIDTH"], c["CAR_HEIGHT"])
        self.masks = {k: (O.blend_mask_for(CAMS[k], *geo) if blend else O.direct_mask(CAMS[k], *geo)) for k in self.cams}
        self.weights = {k: O.blend_weight(m) for k, m in self.masks.items()} if blend else {}
        self.bw, self.bh = c["BEV_WIDTH"], c["BEV_HEIGHT"]
        self.box = mask_box([self.masks[k] for k in self.cams], self.bw, self.bh)

    def vsums(self, frames):
        return np.array([[O.sum_v(f) for f in fs] for fs in frames], np.uint64).reshape(len(frames), len(self.cams))

    def _mask(self, k, img):
        out = np.empty_like(img)
        if self.blend:
            O.lib().orc_weight_mul(O._p(img), O._p(self.weights[k]), img.size, O._p(out))
        else:
            O.lib().orc_mask_select(O._p(img), O._p(self.masks[k]), img.size // 3, O._p(out))
        return out

    def partial(self, frames, all_vsums=None):
        x0, y0, x1, y1 = self.box
        out = []
        for b, fs in enumerate(frames):
            acc = np.zeros((self.bh, self.bw, 3), np.uint8)
            for j, k in enumerate(self.cams):
                img = np.ascontiguousarray(fs[j])
                if self.balance:
                    npx = img.size // 3
                    m = [int(s) / npx for s in all_vsums[b]]
                    v_mean = (m[0] + m[1] + m[2] + m[3]) / 4
                    d = np.empty_like(img)
                    O.lib().orc_luminance_shift(O._p(img), npx, O.lib().orc_round_delta(v_mean - m[k]), O._p(d))
                    img = d
                acc = O.add_sat(acc, self._mask(k, self.cameras[k].raw2bev(img)))
            out.append(acc[y0:y1, x0:x1])
        return np.ascontiguousarray(np.stack(out))

    def combine(self, parts, boxes, car=None):
        batch = parts[0].shape[0]
        out = np.zeros((batch, self.bh, self.bw, 3), np.uint8)
        for b in range(batch):
            for p, (x0, y0, x1, y1) in zip(parts, boxes):
                out[b, y0:y1, x0:x1] = O.add_sat(np.ascontiguousarray(out[b, y0:y1, x0:x1]), np.ascontiguousarray(p[b]))
            if self.balance:
                out[b] = O.color_balance(out[b])
            if car is not None:
                out[b] = O.add_sat(out[b], car)
        return out
