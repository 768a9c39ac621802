#!/usr/bin/env python3
"""Synthetic VP8 stream writer for parity tests (SURVEY.md 8c "stream sources" #2, done on our side of the fence).

Emits LEGAL VP8 frames (RFC 6386 syntax, profile 0) whose content is random but whose FEATURES are chosen by the
caller: SPLITMV in all four partitionings, golden/altref references with sign bias, buffer copies, hidden frames,
segmentation (map + absolute/delta quantiser and loop-filter data), mode/ref loop-filter deltas, sharpness, 1..8
DCT partitions, coefficient / mode / MV probability updates, B_PRED and intra MBs inside inter frames, MVs pointing
far outside the frame, large coefficient categories, odd frame sizes.  The reference encoder never emits most of these
(encode_inter.cc:268, encoder.cc:464-470), so decoder-vs-decoder parity on them needs such streams.

Decoder-vs-decoder parity does not need the stream to look like video.  The writer mirrors the decoder's context
rules (token contexts, b-mode contexts, near-MV census, split-MV contexts) so that what it intends is what decoders
parse; tests check that (intent == oracle parse) as well.  Test tooling only; pure Python, sized for small frames.
"""
import os
import random
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

DC_PRED, V_PRED, H_PRED, TM_PRED, B_PRED, NEARESTMV, NEARMV, ZEROMV, NEWMV, SPLITMV = range(10)
B_DC_PRED, B_TM_PRED, B_VE_PRED, B_HE_PRED, B_LD_PRED, B_RD_PRED, B_VR_PRED, B_VL_PRED, B_HD_PRED, B_HU_PRED, LEFT4X4, ABOVE4X4, ZERO4X4, NEW4X4 = range(14)

KF_Y_MODE_TREE = [-B_PRED, 2, 4, 6, -DC_PRED, -V_PRED, -H_PRED, -TM_PRED]
Y_MODE_TREE = [-DC_PRED, 2, 4, 6, -V_PRED, -H_PRED, -TM_PRED, -B_PRED]
UV_MODE_TREE = [-DC_PRED, 2, -V_PRED, 4, -H_PRED, -TM_PRED]
B_MODE_TREE = [-B_DC_PRED, 2, -B_TM_PRED, 4, -B_VE_PRED, 6, 8, 12, -B_HE_PRED, 10, -B_RD_PRED, -B_VR_PRED,
               -B_LD_PRED, 14, -B_VL_PRED, 16, -B_HD_PRED, -B_HU_PRED]
SMALL_MV_TREE = [2, 8, 4, 6, -0, -1, -2, -3, 10, 12, -4, -5, -6, -7]
MV_REF_TREE = [-ZEROMV, 2, -NEARESTMV, 4, -NEARMV, 6, -NEWMV, -SPLITMV]
SUBMV_REF_TREE = [-LEFT4X4, 2, -ABOVE4X4, 4, -ZERO4X4, -NEW4X4]
SPLIT_MV_TREE = [-3, 2, -2, 4, -0, -1]
SEGMENT_ID_TREE = [2, 4, -0, -1, -2, -3]
ZIGZAG = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]
BAND = [0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7]
SPLIT_LAYOUT = [[0] * 8 + [1] * 8, [0, 0, 1, 1] * 4, [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3], list(range(16))]
SPLIT_COUNT = [2, 2, 4, 16]
CAT = [(7, [165, 145]), (11, [173, 148, 140]), (19, [176, 155, 140, 135]), (35, [180, 157, 141, 134, 130]),
       (67, [254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129])]


def load_tables():
    """Constant tables from our generated header (tools/gen_tables.py output)."""
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "vp8_tables.h")
    text = open(path).read()
    t = {}
    for m in re.finditer(r"vp8o_(\w+)\[[^\]]*\] = \{([^}]*)\}", text):
        body = re.sub(r"/\*.*?\*/", "", m.group(2), flags=re.S)
        t[m.group(1)] = [int(x) for x in body.replace("\n", " ").split(",") if x.strip()]
    return t


T = load_tables()


class BoolEncoder:
    """RFC 6386 section 7.3 boolean entropy encoder."""

    def __init__(self):
        self.out = bytearray(); self.range = 255; self.bottom = 0; self.bit_count = 24

    def _carry(self):
        i = len(self.out) - 1
        while i >= 0 and self.out[i] == 255:
            self.out[i] = 0; i -= 1
        self.out[i] += 1

    def put(self, bit, prob=128):
        split = 1 + (((self.range - 1) * prob) >> 8)
        if bit:
            self.bottom += split; self.range -= split
        else:
            self.range = split
        while self.range < 128:
            self.range <<= 1
            if self.bottom & (1 << 31):
                self._carry()
            self.bottom = (self.bottom << 1) & 0xFFFFFFFF
            self.bit_count -= 1
            if self.bit_count == 0:
                self.out.append(self.bottom >> 24); self.bottom &= (1 << 24) - 1; self.bit_count = 8

    def literal(self, value, bits):
        for i in range(bits - 1, -1, -1):
            self.put((value >> i) & 1)

    def signed(self, value, bits):
        self.literal(abs(value), bits); self.put(1 if value < 0 else 0)

    def flagged_signed(self, value, bits):      # Flagged<Signed<n>>; value None = absent
        if value is None:
            self.put(0)
        else:
            self.put(1); self.signed(value, bits)

    def tree(self, tree, probs, value):
        path = self._path(tree, 0, value)
        for node, bit in path:
            self.put(bit, probs[node >> 1])

    def _path(self, tree, i, value):
        for bit in (0, 1):
            nxt = tree[i + bit]
            if nxt <= 0:
                if -nxt == value:
                    return [(i, bit)]
            else:
                sub = self._path(tree, nxt, value)
                if sub is not None:
                    return [(i, bit)] + sub
        return None

    def finish(self):
        c = self.bit_count; v = self.bottom
        if v & (1 << (32 - c)):
            self._carry()
        v = (v << (c & 7)) & 0xFFFFFFFF; c >>= 3
        while c > 0:
            v = (v << 8) & 0xFFFFFFFF; c -= 1
        for _ in range(4):
            self.out.append((v >> 24) & 0xFF); v = (v << 8) & 0xFFFFFFFF
        return bytes(self.out)


class MBPlan:
    __slots__ = ("y_mode", "uv_mode", "ref", "segment", "skip", "b_modes", "mvs", "partition", "sub_modes", "coeffs", "inter", "flipped")


class SynthStream:
    """Stateful writer: mirrors the decoder's persistent state so successive frames use the right probabilities."""

    def __init__(self, width, height, seed=1):
        self.w, self.h = width, height
        self.mbw, self.mbh = (width + 15) // 16, (height + 15) // 16
        self.rng = random.Random(seed)
        self.coeff_probs = list(T["default_coeff_probs"]); self.y_mode_probs = list(T["default_y_mode_probs"])
        self.uv_mode_probs = list(T["default_uv_mode_probs"]); self.mv_probs = list(T["default_mv_probs"])
        self.seg_enabled = False
        self.frames = []
        self.intent = []     # per frame: list of MBPlan (row-major) for intent-vs-parse checks

    # ---------------------------------------------------------------- header pieces
    def _write_common_header(self, e, p):
        rng = self.rng
        seg = p.get("segmentation")
        if seg is None:
            e.put(0)
        else:
            e.put(1); e.put(1 if seg.get("update_map") else 0)
            data = seg.get("data")
            e.put(1 if data else 0)
            if data:
                e.put(1 if data["absolute"] else 0)
                for v in data["quant"]: e.flagged_signed(v, 7)
                for v in data["lf"]: e.flagged_signed(v, 6)
            if seg.get("update_map"):
                for v in seg["tree_probs"]:
                    if v is None: e.put(0)
                    else: e.put(1); e.literal(v, 8)
        e.put(0)                                  # filter_type: normal
        e.literal(p["lf_level"], 6); e.literal(p["sharpness"], 3)
        adj = p.get("lf_deltas")
        if adj is None:
            e.put(0)
        else:
            e.put(1)
            if adj.get("update"):
                e.put(1)
                for v in adj["ref"]: e.flagged_signed(v, 6)
                for v in adj["mode"]: e.flagged_signed(v, 6)
            else:
                e.put(0)
        e.literal(p["log2_parts"], 2)
        e.literal(p["q_index"], 7)
        for v in p["q_deltas"]: e.flagged_signed(v, 4)

    def _write_coeff_prob_updates(self, e, frame_probs, n_updates):
        upd = T["coeff_update_probs"]
        chosen = {}
        for _ in range(n_updates):
            chosen[self.rng.randrange(1056)] = self.rng.randint(1, 255)
        for i in range(1056):
            if i in chosen:
                e.put(1, upd[i]); e.literal(chosen[i], 8); frame_probs[i] = chosen[i]
            else:
                e.put(0, upd[i])

    # ---------------------------------------------------------------- tokens
    def _write_block(self, e, probs, btype, ctx, coeffs, first):
        """coeffs: 16 values in de-zigzagged positions.  Returns has_nonzero."""
        zz = [coeffs[ZIGZAG[i]] for i in range(16)]
        last = -1
        for i in range(first, 16):
            if zz[i] != 0: last = i
        nonzero = False
        last_was_zero = False
        i = first
        while i < 16:
            p = probs[((btype * 8 + BAND[i]) * 3 + ctx) * 11:((btype * 8 + BAND[i]) * 3 + ctx) * 11 + 11]
            if not last_was_zero:
                if i > last:
                    e.put(0, p[0]); return nonzero       # EOB
                e.put(1, p[0])
            v = zz[i]
            if v == 0:
                e.put(0, p[1]); last_was_zero = True; ctx = 0; i += 1; continue
            e.put(1, p[1]); last_was_zero = False; nonzero = True
            a = abs(v)
            if a == 1:
                e.put(0, p[2]); ctx = 1
            else:
                e.put(1, p[2]); ctx = 2
                if a <= 4:
                    e.put(0, p[3])
                    if a == 2: e.put(0, p[4])
                    else: e.put(1, p[4]); e.put(a - 3, p[5])
                elif a <= 10:
                    e.put(1, p[3]); e.put(0, p[6])
                    if a <= 6: e.put(0, p[7]); e.put(a - 5, 159)
                    else: e.put(1, p[7]); self._cat(e, 0, a)
                else:
                    e.put(1, p[3]); e.put(1, p[6])
                    if a < 35:
                        e.put(0, p[8])
                        if a < 19: e.put(0, p[9]); self._cat(e, 1, a)
                        else: e.put(1, p[9]); self._cat(e, 2, a)
                    else:
                        e.put(1, p[8])
                        if a < 67: e.put(0, p[10]); self._cat(e, 3, a)
                        else: e.put(1, p[10]); self._cat(e, 4, a)
            e.put(1 if v < 0 else 0, 128)
            i += 1
        return nonzero

    @staticmethod
    def _cat(e, cat, a):
        base, probs = CAT[cat]
        inc = a - base
        for k, pr in enumerate(probs):
            e.put((inc >> (len(probs) - 1 - k)) & 1, pr)

    def _rand_coeffs(self, density, first, big):
        rng = self.rng
        c = [0] * 16
        if rng.random() > density:
            return c
        n = rng.choice([1, 1, 2, 3, 5, 8, 16])
        for _ in range(n):
            pos = rng.randrange(first, 16)
            r = rng.random()
            if r < 0.55: mag = 1
            elif r < 0.8: mag = rng.randint(2, 4)
            elif r < 0.93: mag = rng.randint(5, 34)
            elif r < 0.99 or not big: mag = rng.randint(35, 66)
            else: mag = rng.randint(67, 67 + 2047)
            c[ZIGZAG[pos]] = mag if rng.random() < 0.5 else -mag
        return c

    # ---------------------------------------------------------------- motion vectors
    def _write_mv_component(self, e, p, v):
        a = abs(v) >> 1
        if a < 8:
            e.put(0, p[0]); e.tree(SMALL_MV_TREE, p[2:9], a)
        else:
            e.put(1, p[0])
            for i in range(3): e.put((a >> i) & 1, p[9 + i])
            for i in range(9, 3, -1): e.put((a >> i) & 1, p[9 + i])
            if a & 0xFFF0: e.put((a >> 3) & 1, p[9 + 3])
        if a: e.put(1 if v < 0 else 0, p[1])

    def _write_mv(self, e, probs, dx, dy):
        self._write_mv_component(e, probs[0:19], dy); self._write_mv_component(e, probs[19:38], dx)

    def _clamp_mv(self, mv, col, row):
        to_left = -((col * 16) << 3) - 128; to_right = (((self.mbw - 1 - col) * 16) << 3) + 128
        to_top = -((row * 16) << 3) - 128; to_bottom = (((self.mbh - 1 - row) * 16) << 3) + 128
        return (min(max(mv[0], to_left), to_right), min(max(mv[1], to_top), to_bottom))

    def _census(self, plans, col, row, flipped):
        score = [0, 0, 0, 0]; cand = [(0, 0)] * 4; idx = 0; split = 0
        for (c, r, w) in ((col, row - 1, 2), (col - 1, row, 2), (col - 1, row - 1, 1)):
            if c < 0 or r < 0: continue
            nb = plans[r * self.mbw + c]
            if not nb.inter: continue
            mv = nb.mvs[15]
            if nb.flipped != flipped: mv = (-mv[0], -mv[1])
            if mv == (0, 0): score[0] += w
            else:
                if mv != cand[idx]:
                    idx += 1; cand[idx] = mv
                score[idx] += w
            if nb.y_mode == SPLITMV: split += w
        if score[3] and cand[idx] == cand[1]: score[1] += score[3]
        if score[2] > score[1]:
            score[1], score[2] = score[2], score[1]; cand[1], cand[2] = cand[2], cand[1]
        if score[1] >= score[0]: cand[0] = cand[1]
        return cand, [score[0], score[1], score[2], split]

    # ---------------------------------------------------------------- one frame
    def frame(self, key=False, show=True, lf_level=0, sharpness=0, q_index=40, q_deltas=(None,) * 5, log2_parts=0,
              segmentation=None, lf_deltas=None, refresh_entropy=True, coeff_updates=0, skip_prob=None,
              density=0.3, big_coeffs=False, prob_inter=200, prob_last=128, prob_golden=128, inter_modes=None,
              intra_bpred=0.3, mv_range=40, refresh_golden=False, refresh_alt=False, copy_golden=0, copy_alt=0,
              sign_bias_golden=False, sign_bias_alt=False, refresh_last=True, update_mode_probs=False, mv_prob_updates=0,
              skip_rate=0.8):
        rng = self.rng
        mbw, mbh = self.mbw, self.mbh
        e = BoolEncoder()
        p = dict(lf_level=lf_level, sharpness=sharpness, q_index=q_index, q_deltas=list(q_deltas), log2_parts=log2_parts,
                 segmentation=segmentation, lf_deltas=lf_deltas)
        if key:
            e.put(0); e.put(0)                    # color_space, clamping_type
            self.coeff_probs = list(T["default_coeff_probs"]); self.y_mode_probs = list(T["default_y_mode_probs"])
            self.uv_mode_probs = list(T["default_uv_mode_probs"]); self.mv_probs = list(T["default_mv_probs"])
            self.seg_enabled = False
        self._write_common_header(e, p)
        fp_coeff = list(self.coeff_probs); fp_y = list(self.y_mode_probs); fp_uv = list(self.uv_mode_probs); fp_mv = list(self.mv_probs)
        if key:
            e.put(1 if refresh_entropy else 0)
        else:
            e.put(1 if refresh_golden else 0); e.put(1 if refresh_alt else 0)
            if not refresh_golden: e.literal(copy_golden, 2)
            if not refresh_alt: e.literal(copy_alt, 2)
            e.put(1 if sign_bias_golden else 0); e.put(1 if sign_bias_alt else 0)
            e.put(1 if refresh_entropy else 0); e.put(1 if refresh_last else 0)
        self._write_coeff_prob_updates(e, fp_coeff, coeff_updates)
        if skip_prob is None:
            e.put(0)
        else:
            e.put(1); e.literal(skip_prob, 8)
        if not key:
            e.literal(prob_inter, 8); e.literal(prob_last, 8); e.literal(prob_golden, 8)
            if update_mode_probs:
                e.put(1); fp_y = [rng.randint(1, 255) for _ in range(4)]
                for v in fp_y: e.literal(v, 8)
                e.put(1); fp_uv = [rng.randint(1, 255) for _ in range(3)]
                for v in fp_uv: e.literal(v, 8)
            else:
                e.put(0); e.put(0)
            chosen = {rng.randrange(38): rng.randint(0, 127) for _ in range(mv_prob_updates)}
            for i in range(38):
                if i in chosen:
                    e.put(1, T["mv_update_probs"][i]); e.literal(chosen[i], 7); fp_mv[i] = (chosen[i] << 1) if chosen[i] else 1
                else:
                    e.put(0, T["mv_update_probs"][i])
        if refresh_entropy:
            self.coeff_probs, self.y_mode_probs, self.uv_mode_probs, self.mv_probs = list(fp_coeff), list(fp_y), list(fp_uv), list(fp_mv)
        seg_on = segmentation is not None
        update_map = seg_on and segmentation.get("update_map")
        seg_tree_probs = [255 if v is None else v for v in segmentation["tree_probs"]] if update_map else None

        # ---- plan + write macroblock headers ----
        nparts = 1 << log2_parts
        parts = [BoolEncoder() for _ in range(nparts)]
        plans = []
        above_nz = [[0] * 9 for _ in range(mbw)]
        modes = inter_modes or [NEARESTMV, NEARMV, ZEROMV, NEWMV, SPLITMV]
        for row in range(mbh):
            left_nz = [0] * 9
            for col in range(mbw):
                mb = MBPlan(); plans.append(mb)
                mb.segment = rng.randrange(4) if update_map else 0
                if update_map: e.tree(SEGMENT_ID_TREE, seg_tree_probs, mb.segment)
                mb.inter = (not key) and rng.random() < prob_inter / 256.0
                mb.flipped = False; mb.mvs = [(0, 0)] * 16; mb.b_modes = [0] * 16; mb.partition = 0; mb.ref = 0
                # modes are chosen first (they decide whether a Y2 block exists), the skip flag is written before them
                if not mb.inter:
                    mb.y_mode = B_PRED if rng.random() < intra_bpred else rng.choice([DC_PRED, V_PRED, H_PRED, TM_PRED])
                    mb.uv_mode = rng.choice([DC_PRED, V_PRED, H_PRED, TM_PRED])
                else:
                    mb.ref = 1      # reference choice follows the signalled probabilities loosely
                    if rng.random() < (256 - prob_last) / 256.0:
                        mb.ref = 3 if rng.random() < (256 - prob_golden) / 256.0 else 2
                    mb.flipped = (mb.ref == 2 and sign_bias_golden) or (mb.ref == 3 and sign_bias_alt)
                    mb.y_mode = rng.choice(modes)
                has_y2 = mb.y_mode not in (B_PRED, SPLITMV)
                coeffs = [[0] * 16 for _ in range(25)]
                any_nz = False
                if rng.random() < 0.85:
                    for b in range(24):
                        coeffs[b] = self._rand_coeffs(density, 1 if (has_y2 and b < 16) else 0, big_coeffs)
                    if has_y2: coeffs[24] = self._rand_coeffs(min(1.0, density * 2), 0, big_coeffs)
                    any_nz = any(any(c) for c in coeffs)
                mb.skip = False
                if skip_prob is not None:
                    mb.skip = (not any_nz) and rng.random() < skip_rate
                    e.put(1 if mb.skip else 0, skip_prob)
                mb.coeffs = coeffs
                if not key:
                    e.put(1 if mb.inter else 0, prob_inter)
                    if mb.inter:
                        e.put(0 if mb.ref == 1 else 1, prob_last)
                        if mb.ref != 1: e.put(1 if mb.ref == 3 else 0, prob_golden)
                if not mb.inter:
                    if key: e.tree(KF_Y_MODE_TREE, T["kf_y_mode_probs"], mb.y_mode)
                    else: e.tree(Y_MODE_TREE, fp_y, mb.y_mode)
                    if mb.y_mode == B_PRED:
                        for b in range(16):
                            m = rng.randrange(10); mb.b_modes[b] = m
                            if key:
                                am = lm = B_DC_PRED
                                if b >= 4: am = mb.b_modes[b - 4]
                                elif row > 0: am = plans[(row - 1) * mbw + col].b_modes[b + 12]
                                if b & 3: lm = mb.b_modes[b - 1]
                                elif col > 0: lm = plans[row * mbw + col - 1].b_modes[b + 3]
                                base = (am * 10 + lm) * 9
                                e.tree(B_MODE_TREE, T["kf_b_mode_probs"][base:base + 9], m)
                            else:
                                e.tree(B_MODE_TREE, T["b_mode_probs"], m)
                    else:
                        mb.b_modes = [{DC_PRED: B_DC_PRED, V_PRED: B_VE_PRED, H_PRED: B_HE_PRED, TM_PRED: B_TM_PRED}[mb.y_mode]] * 16
                    if key: e.tree(UV_MODE_TREE, T["kf_uv_mode_probs"], mb.uv_mode)
                    else: e.tree(UV_MODE_TREE, fp_uv, mb.uv_mode)
                else:
                    cand, ctx = self._census(plans, col, row, mb.flipped)
                    mode_probs = [T["mv_counts_to_probs"][ctx[i] * 4 + i] for i in range(4)]
                    e.tree(MV_REF_TREE, mode_probs, mb.y_mode)
                    best = self._clamp_mv(cand[0], col, row)

                    def new_target():
                        # mostly modest vectors, sometimes far outside the frame (edge clamping path)
                        if rng.random() < 0.15:
                            return (rng.randrange(-700, 701) * 2, rng.randrange(-700, 701) * 2)
                        return (rng.randrange(-mv_range, mv_range + 1) * 2, rng.randrange(-mv_range, mv_range + 1) * 2)

                    def write_new(target):
                        dx = max(-2046, min(2046, target[0] - best[0])); dy = max(-2046, min(2046, target[1] - best[1]))
                        self._write_mv(e, fp_mv, dx, dy)
                        return (best[0] + dx, best[1] + dy)

                    if mb.y_mode == NEARESTMV: base = self._clamp_mv(cand[1], col, row)
                    elif mb.y_mode == NEARMV: base = self._clamp_mv(cand[2], col, row)
                    elif mb.y_mode == ZEROMV: base = (0, 0)
                    elif mb.y_mode == NEWMV: base = write_new(new_target())
                    else:
                        mb.partition = rng.randrange(4)
                        e.tree(SPLIT_MV_TREE, T["split_mv_probs"], mb.partition)
                        layout = SPLIT_LAYOUT[mb.partition]
                        mvs = [(0, 0)] * 16
                        for part in range(SPLIT_COUNT[mb.partition]):
                            b = layout.index(part)
                            lmv = amv = (0, 0)
                            if b & 3: lmv = mvs[b - 1]
                            elif col > 0 and plans[row * mbw + col - 1].inter: lmv = plans[row * mbw + col - 1].mvs[b + 3]
                            if b >= 4: amv = mvs[b - 4]
                            elif row > 0 and plans[(row - 1) * mbw + col].inter: amv = plans[(row - 1) * mbw + col].mvs[b + 12]
                            if lmv == amv: sctx = 4 if lmv == (0, 0) else 3
                            elif amv == (0, 0): sctx = 2
                            elif lmv == (0, 0): sctx = 1
                            else: sctx = 0
                            sm = rng.choice([LEFT4X4, ABOVE4X4, ZERO4X4, NEW4X4, NEW4X4])
                            e.tree(SUBMV_REF_TREE, T["submv_ref_probs"][sctx * 3:sctx * 3 + 3], sm)
                            if sm == LEFT4X4: m = lmv
                            elif sm == ABOVE4X4: m = amv
                            elif sm == ZERO4X4: m = (0, 0)
                            else: m = write_new(new_target())
                            for k in range(16):
                                if layout[k] == part: mvs[k] = m
                        mb.mvs = mvs
                        base = None
                    if base is not None:
                        mb.mvs = [base] * 16
                # ---- tokens into the row's DCT partition ----
                te = parts[row % nparts]
                anz = above_nz[col]
                if mb.skip:
                    for k in range(8): anz[k] = 0; left_nz[k] = 0
                    if has_y2: anz[8] = 0; left_nz[8] = 0
                else:
                    if has_y2:
                        nz = self._write_block(te, fp_coeff, 1, anz[8] + left_nz[8], coeffs[24], 0)
                        anz[8] = left_nz[8] = int(nz)
                    for b in range(16):
                        nz = self._write_block(te, fp_coeff, 0 if has_y2 else 3, anz[b & 3] + left_nz[b >> 2], coeffs[b], 1 if has_y2 else 0)
                        anz[b & 3] = left_nz[b >> 2] = int(nz)
                    for pl in range(2):
                        for b in range(4):
                            ia, il = 4 + pl * 2 + (b & 1), 4 + pl * 2 + (b >> 1)
                            nz = self._write_block(te, fp_coeff, 2, anz[ia] + left_nz[il], coeffs[16 + pl * 4 + b], 0)
                            anz[ia] = left_nz[il] = int(nz)
        first = e.finish()
        part_bytes = [pe.finish() for pe in parts]
        # ---- frame tag (RFC 6386 9.1), key-frame start code + dimensions ----
        tag = (0 if key else 1) | (0 << 1) | ((1 if show else 0) << 4) | (len(first) << 5)
        out = bytearray(struct.pack("<I", tag)[:3])
        if key:
            out += b"\x9d\x01\x2a" + struct.pack("<HH", self.w, self.h)
        out += first
        for pb in part_bytes[:-1]:
            out += struct.pack("<I", len(pb))[:3]
        for pb in part_bytes:
            out += pb
        self.frames.append(bytes(out)); self.intent.append(plans)
        return bytes(out)


def feature_stream(width, height, seed, nframes=8):
    """A stream that walks through the features listed in the module docstring."""
    s = SynthStream(width, height, seed)
    rng = random.Random(seed * 7919 + 1)
    s.frame(key=True, lf_level=rng.randint(1, 63), sharpness=rng.randrange(8), q_index=rng.randrange(128),
            q_deltas=[rng.choice([None, rng.randint(-15, 15)]) for _ in range(5)], log2_parts=rng.randrange(4),
            segmentation=dict(update_map=True, tree_probs=[rng.choice([None, rng.randint(1, 255)]) for _ in range(3)],
                              data=dict(absolute=rng.random() < 0.5, quant=[rng.choice([None, rng.randint(-60, 100)]) for _ in range(4)],
                                        lf=[rng.choice([None, rng.randint(-30, 50)]) for _ in range(4)])),
            lf_deltas=dict(update=True, ref=[rng.choice([None, rng.randint(-20, 20)]) for _ in range(4)],
                           mode=[rng.choice([None, rng.randint(-20, 20)]) for _ in range(4)]),
            coeff_updates=rng.randrange(30), skip_prob=rng.choice([None, rng.randint(1, 255)]), density=rng.random() * 0.6,
            big_coeffs=True, intra_bpred=0.5, refresh_entropy=rng.random() < 0.7)
    for i in range(1, nframes):
        seg = None
        r = rng.random()
        if r < 0.35:
            seg = dict(update_map=rng.random() < 0.5, tree_probs=[rng.choice([None, rng.randint(1, 255)]) for _ in range(3)],
                       data=None if rng.random() < 0.4 else dict(absolute=rng.random() < 0.5,
                                                                quant=[rng.choice([None, rng.randint(-100, 120)]) for _ in range(4)],
                                                                lf=[rng.choice([None, rng.randint(-63, 63)]) for _ in range(4)]))
        elif r < 0.6:
            seg = dict(update_map=False, tree_probs=[None] * 3, data=None)
        lfd = None
        r = rng.random()
        if r < 0.4:
            lfd = dict(update=True, ref=[rng.choice([None, rng.randint(-30, 30)]) for _ in range(4)],
                       mode=[rng.choice([None, rng.randint(-30, 30)]) for _ in range(4)])
        elif r < 0.6:
            lfd = dict(update=False)
        rg, ra = rng.random() < 0.25, rng.random() < 0.25
        s.frame(key=False, show=rng.random() < 0.85 or i == nframes - 1, lf_level=rng.choice([0, rng.randint(1, 63), rng.randint(1, 63)]),
                sharpness=rng.randrange(8), q_index=rng.randrange(128), q_deltas=[rng.choice([None, rng.randint(-15, 15)]) for _ in range(5)],
                log2_parts=rng.randrange(4), segmentation=seg, lf_deltas=lfd, refresh_entropy=rng.random() < 0.6,
                coeff_updates=rng.randrange(20), skip_prob=rng.choice([None, rng.randint(1, 255)]), density=rng.random() * 0.5,
                big_coeffs=rng.random() < 0.3, prob_inter=rng.choice([255, 230, 128]), prob_last=rng.choice([255, 128, 60]),
                prob_golden=rng.choice([200, 128, 30]), intra_bpred=0.5, mv_range=rng.choice([4, 40, 200]),
                refresh_golden=rg, refresh_alt=ra, copy_golden=rng.randrange(3), copy_alt=rng.randrange(3),
                sign_bias_golden=rng.random() < 0.5, sign_bias_alt=rng.random() < 0.5, refresh_last=rng.random() < 0.8,
                update_mode_probs=rng.random() < 0.3, mv_prob_updates=rng.randrange(6))
    return s


def perf_stream(width, height, seed, nframes=12):
    """The benchmark's "realistic inter" workload (bench.py --config 1080p_inter_lf_subpel): 1 key + nframes-1 inter frames,
    high entropy, loop filter 24, four DCT partitions; inter macroblocks ~97 %, of which ~15 % SPLITMV, ~45 % NEWMV with
    quarter-pel vectors (7 of 8 have a fractional part), the rest NEAREST / NEAR / ZERO; LAST / GOLDEN / ALTREF all in use,
    golden and altref refreshed now and then.  The reference encoder emits none of this (full-pel vectors, LAST only, no
    SPLITMV, one partition: SURVEY.md 8c)."""
    s = SynthStream(width, height, seed)
    rng = random.Random(seed * 104729 + 7)
    s.frame(key=True, lf_level=24, sharpness=0, q_index=20, log2_parts=2, skip_prob=40, density=0.5, intra_bpred=0.5)
    modes = [NEARESTMV] * 4 + [NEARMV] * 2 + [ZEROMV] * 2 + [NEWMV] * 9 + [SPLITMV] * 3
    for i in range(1, nframes):
        s.frame(key=False, show=True, lf_level=24, sharpness=0, q_index=20, log2_parts=2, skip_prob=60, density=0.45,
                prob_inter=248, prob_last=150, prob_golden=128, inter_modes=modes, intra_bpred=0.3, mv_range=24,
                refresh_golden=i % 5 == 0, refresh_alt=i % 7 == 0, copy_golden=0, copy_alt=0,
                sign_bias_golden=False, sign_bias_alt=i % 2 == 0, refresh_last=True, mv_prob_updates=2 if i == 1 else 0,
                refresh_entropy=True, coeff_updates=4 if i == 1 else 0)
    return s


if __name__ == "__main__":
    import argparse
    from ivf_io import write_ivf
    ap = argparse.ArgumentParser()
    ap.add_argument("out"); ap.add_argument("--width", type=int, default=96); ap.add_argument("--height", type=int, default=80)
    ap.add_argument("--frames", type=int, default=8); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    st = feature_stream(a.width, a.height, a.seed, a.frames)
    write_ivf(a.out, a.width, a.height, st.frames)
    print(a.out, [len(f) for f in st.frames])
