#!/usr/bin/env python3
"""Generate xbitops_amd/csrc/xbit_unpack_gen.h: field-pair extraction plans for 3/5/6/7-bit weights.

A k-block (32 fields of b bits) is b words.  Fields are paired 8 apart — (k, k+8) inside each
group of 16 — because field k+8 starts exactly b BYTES after field k: two byte-aligned 16-bit
windows b bytes apart show the two fields of a pair at the SAME bit offset.  One `v_perm_b32`
builds X = [window_lo | window_hi]; then, exactly as for 2/4/8 bit, each pair costs one
`v_and_or_b32` ((X >> s) & (mask << e) | 0x64006400 = (1024 + q_lo*2^e, 1024 + q_hi*2^e)), with one
shift shared by the fields whose offset exceeds the 10-bit mantissa.

For every (bits, pair p) the generator emits one C++ expression; common sub-expressions (the perm,
the shift) are identical text and are merged by the compiler.  It also emits `pair_exp(p)` (= e).
The plan is verified here by simulating the emitted operations on random words against a plain
bit-stream unpack before the header is written.

    python tools/gen_unpack.py            # rewrites the header (and checks it)
"""
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "xbitops_amd", "csrc", "xbit_unpack_gen.h")


def plan_for(bits):
    """Returns list over pairs p (0..15) of dict(build=..., shift=s, e=e) for the 32-field k-block."""
    max_e = 10 - bits
    pairs = [None] * 16
    for d in range(2):  # group of 16 fields
        todo = list(range(8))
        while todo:
            i0 = todo[0]
            bit0 = (16 * d + i0) * bits
            beta_a = bit0 // 8                 # byte offset of the low window in the k-block stream
            beta_b = beta_a + bits             # field k+8 is `bits` bytes further
            t_a = 8 * beta_a
            covered = [i for i in todo if (16 * d + i) * bits >= t_a and (16 * d + i) * bits + bits <= t_a + 16]
            assert covered and covered[0] == i0
            # how X is built
            words = sorted({b // 4 for b in (beta_a, beta_a + 1, beta_b, beta_b + 1)})
            if len(words) <= 2:
                u = words[0]
                v = words[-1]

                def selbyte(b, u=u):
                    return (b % 4) if b // 4 == u else 4 + (b % 4)

                sel = selbyte(beta_a) | (selbyte(beta_a + 1) << 8) | (selbyte(beta_b) << 16) | (selbyte(beta_b + 1) << 24)
                build = ("perm", v, u, sel)    # __builtin_amdgcn_perm(w[v], w[u], sel)
            else:
                # more than two words: a window that straddles two words is first funnel-shifted into
                # one register; a window inside one word is addressed by byte selection as before
                def win(beta):
                    if beta % 4 <= 2:
                        return ("word", beta // 4), beta % 4          # (source, first byte inside it)
                    return ("align", beta // 4 + 1, beta // 4, 24), 0  # bytes 3|0 of two words -> bytes 0,1

                (src_a, ba), (src_b, bb) = win(beta_a), win(beta_b)
                sel = ba | ((ba + 1) << 8) | ((4 + bb) << 16) | ((5 + bb) << 24)
                build = ("perm2", src_b, src_a, sel)
            # group the covered fields by shared shift
            offs = {i: (16 * d + i) * bits - t_a for i in covered}
            rest = list(covered)
            while rest:
                s = offs[rest[0]]
                grp = [i for i in rest if 0 <= offs[i] - s <= max_e]
                for i in grp:
                    pairs[8 * d + i] = dict(build=build, shift=s, e=offs[i] - s)
                rest = [i for i in rest if i not in grp]
            todo = [i for i in todo if i not in covered]
    return pairs


def expr_build(b):
    kind = b[0]
    if kind == "perm":
        return "__builtin_amdgcn_perm(w[%d], w[%d], 0x%08xu)" % (b[1], b[2], b[3])

    def win(wn):
        if wn[0] == "word":
            return "w[%d]" % wn[1]
        return "__builtin_amdgcn_alignbit(w[%d], w[%d], %d)" % (wn[1], wn[2], wn[3])

    return "__builtin_amdgcn_perm(%s, %s, 0x%08xu)" % (win(b[1]), win(b[2]), b[3])


def simulate(bits, pairs, trials=200):
    rng = random.Random(bits)
    for _ in range(trials):
        w = [rng.getrandbits(32) for _ in range(bits)]
        stream = sum(x << (32 * i) for i, x in enumerate(w))
        field = lambda k: (stream >> (k * bits)) & ((1 << bits) - 1)

        def perm(s0, s1, sel):
            src = [(s1 >> (8 * i)) & 0xFF for i in range(4)] + [(s0 >> (8 * i)) & 0xFF for i in range(4)]
            return sum(src[(sel >> (8 * i)) & 0xFF] << (8 * i) for i in range(4))

        def win(wn):
            if wn[0] == "word":
                return w[wn[1]]
            return (((w[wn[1]] << 32) | w[wn[2]]) >> wn[3]) & 0xFFFFFFFF

        for p, pl in enumerate(pairs):
            b = pl["build"]
            x = perm(w[b[1]], w[b[2]], b[3]) if b[0] == "perm" else perm(win(b[1]), win(b[2]), b[3])
            q = ((x >> pl["shift"]) & (((1 << bits) - 1) << pl["e"]) * 0x00010001) | 0x64006400
            klo = 16 * (p // 8) + (p % 8)
            lo, hi = q & 0xFFFF, q >> 16
            assert lo == 0x6400 + (field(klo) << pl["e"]), (bits, p)
            assert hi == 0x6400 + (field(klo + 8) << pl["e"]), (bits, p)
            assert pl["e"] + bits <= 10
            # subnormal form: the same fields moved to mantissa bits [E, E + bits) with everything else cleared
            E = 10 - bits
            t = pl["shift"] + pl["e"] - E
            assert -E <= t <= 6
            y = ((x >> t) if t >= 0 else (x << -t)) & 0xFFFFFFFF
            qs = y & ((((1 << bits) - 1) << E) * 0x00010001)
            assert qs & 0xFFFF == field(klo) << E and qs >> 16 == field(klo + 8) << E, (bits, p)


def main():
    lines = [
        "// xbit_unpack_gen.h — GENERATED by tools/gen_unpack.py; do not edit.",
        "//",
        "// Field-pair extraction for 3/5/6/7-bit k-blocks: pair p = (k, k+8) with k = 16*(p/8) + p%8.",
        "// Each expression is (X >> s) & (mask << e) | magic with X = one v_perm_b32 of two byte-aligned",
        "// 16-bit windows `bits` bytes apart (see the generator for the derivation); identical",
        "// sub-expressions are merged by the compiler.  `sub_pair` is the same pair with the fields moved to",
        "// a fixed mantissa position and nothing spliced in (fp16 subnormals q * 2^(10-BITS-24)): one full-rate",
        "// shift + one v_and_b32 per pair on top of the shared perm.  Verified against a bit-stream unpack at",
        "// generation time.",
        "#pragma once",
        "",
        "#include <stdint.h>",
        "",
        "namespace xbit {",
        "",
        "template <int BITS>",
        "struct OddUnpack;",
        "",
    ]
    total_cost = {}
    for bits in (3, 5, 6, 7):
        pairs = plan_for(bits)
        assert all(p is not None for p in pairs)
        simulate(bits, pairs)
        builds = {expr_build(p["build"]) for p in pairs}
        shifts = {(expr_build(p["build"]), p["shift"]) for p in pairs if p["shift"]}
        extra = sum(2 if p["build"][0] == "perm2" and any(wn[0] != "word" for wn in (p["build"][1], p["build"][2])) else 0
                    for p in {id(q["build"]): q for q in pairs}.values())
        total_cost[bits] = len(builds) + len(shifts) + 16
        lines.append("// %d-bit: %d window builds + %d shifts + 16 and_or per 32 fields" % (bits, len(builds), len(shifts)))
        lines.append("template <>")
        lines.append("struct OddUnpack<%d> {" % bits)
        lines.append("  static constexpr int kExp[16] = {%s};" % ", ".join(str(p["e"]) for p in pairs))
        lines.append("  template <int P, typename Consts>")
        lines.append("  __device__ __forceinline__ static uint32_t pair(const uint32_t (&w)[%d], const Consts& k) {" % bits)
        for p, pl in enumerate(pairs):
            x = expr_build(pl["build"])
            if pl["shift"]:
                x = "(%s >> %d)" % (x, pl["shift"])
            lines.append("    %sif constexpr (P == %d) return (%s & k.mask[%d]) | k.magic;" % ("" if p == 0 else "else ", p, x, pl["e"]))
        lines.append("    else return 0u;")
        lines.append("  }")
        lines.append("  // subnormal form: both fields at mantissa bits [10 - BITS, 10), all other bits 0 (mask = field mask << (10 - BITS), both halves)")
        lines.append("  template <int P>")
        lines.append("  __device__ __forceinline__ static uint32_t sub_pair(const uint32_t (&w)[%d], uint32_t mask) {" % bits)
        for p, pl in enumerate(pairs):
            x = expr_build(pl["build"])
            t = pl["shift"] + pl["e"] - (10 - bits)
            if t > 0:
                x = "(%s >> %d)" % (x, t)
            elif t < 0:
                x = "(%s << %d)" % (x, -t)
            lines.append("    %sif constexpr (P == %d) return %s & mask;" % ("" if p == 0 else "else ", p, x))
        lines.append("    else return 0u;")
        lines.append("  }")
        lines.append("};")
        lines.append("")
    lines.append("}  // namespace xbit")
    with open(OUT, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("wrote", OUT, "ops per 32 fields:", total_cost)


if __name__ == "__main__":
    main()
