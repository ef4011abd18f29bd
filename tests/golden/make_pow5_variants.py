"""Known-answer vectors for the two readings of  x ^ (5 :: Int)  in fresnel (src/Material.hs:26-29).

    python tests/golden/make_pow5_variants.py        # writes tests/golden/pow5_variants.json

Plain IEEE-754 double arithmetic in Python (no oracle, no kernel): for each (ior, cos0) the Fresnel factor
r + (1 - r) * (1 - cos0)^5 with the power as
  flat  : x*x*x*x*x            -- what GHC's RULES "^5/Int" (GHC.Real) make of a literal Int exponent under -O; the
                                   reference builds with -O2 (build.sh:1).  THE DEFAULT of oracle and kernels.
  sqmul : ((x*x)*(x*x))*x      -- GHC.Real's (^) as defined (square-and-multiply), what an un-optimised build runs.
Only arguments on which the two differ are kept.  Neither column comes from GHC: there is none in the build image.  One
Fresnel value printed by a real `ghc -O2` build -- e.g.  `printf "%a" (fresnel (Mirror 1.9) 0.3)`-style hex output of
`r0 1 1.9 + (1 - r0 1 1.9) * (1 - 0.3) ^ (5 :: Int)` -- against this file decides which column the reference computes; if it
is `sqmul`, flip the default in oracle/rayhs_oracle.c (g_pow5_sqmul), oracle/second/rayhs_second.py
(POW5_SQUARE_AND_MULTIPLY) and abi.cpp (dp.flags_dev) and regenerate the goldens.
"""
import json
import os
import random


def r0(n1, n2):
    q = (n1 - n2) / (n1 + n2)
    return q * q


def fresnel(ior, cos0, sqmul):
    r = r0(1.0, ior)
    x = 1 - cos0
    p5 = ((x * x) * (x * x)) * x if sqmul else x * x * x * x * x
    return r + (1 - r) * p5


def main():
    rng = random.Random(5237)  # the GHC ticket that introduced the rules
    rows = []
    fixed = [(1.9, 0.3), (1.5, 0.25), (1.33, 0.7), (1.9, -0.5), (2.4, 0.05)]
    cands = fixed + [(rng.choice([1.33, 1.5, 1.9, 2.4]), rng.uniform(-1, 1)) for _ in range(400)]
    for ior, c in cands:
        a, b = fresnel(ior, c, False), fresnel(ior, c, True)
        if a != b and len(rows) < 24:
            rows.append({"ior": ior.hex(), "cos0": c.hex(), "flat": a.hex(), "sqmul": b.hex()})
    out = {"default": "flat", "note": "see make_pow5_variants.py; neither column was produced by GHC", "vectors": rows}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pow5_variants.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(len(rows), "vectors ->", path)


if __name__ == "__main__":
    main()
