#!/usr/bin/env python3
"""Turn the reference's own known-answer data for the MSM path into small fixtures under tests/golden/.

Run HERE (the container that has /root/reference); the fixtures are committed because
/root/reference does not exist on the GPU box. Nothing else in the repo reads /root/reference.

Sources (data files, not code):
  curves/bls12_381/src/curves/tests/g1_uncompressed_valid_test_vectors.dat   k*G1, k = 0..999
  curves/bls12_381/src/curves/tests/g2_uncompressed_valid_test_vectors.dat   k*G2, k = 0..999
      (checked by the reference at curves/bls12_381/src/curves/tests/mod.rs:69-123; zkcrypto
       big-endian encoding per curves/bls12_381/src/curves/util.rs:10-90 and g2.rs serialize:
       G1 = x|y, G2 = x.c1|x.c0|y.c1|y.c0, 48-byte big-endian canonical field elements,
       flag bits in the top 3 bits of byte 0, entry 0 = infinity)
  curves/bls12_377/src/curves/tests/*G2*.json   RFC 9380 hash-to-curve vectors: points P, Q0, Q1
      on BLS12-377 G2 as "c0,c1" big-endian hex (parsed by test-templates/src/h2c/mod.rs:74-83)

Output (numpy .npz, canonical i.e. NON-Montgomery little-endian u64 limbs):
  tests/golden/bls12_381_g1_multiples.npz   xy[1000, 2, 6], infinity[1000]
  tests/golden/bls12_381_g2_multiples.npz   xy[1000, 2, 2, 6] (coordinate, c0/c1, limbs), infinity[1000]
  tests/golden/bls12_377_g2_h2c_points.npz  xy[m, 2, 2, 6]
"""
import glob, json, os, sys
import numpy as np

REF = "/root/reference/curves"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def limbs(x, n=6):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def parse_multiples(path, fe_per_coord):
    raw = open(path, "rb").read()
    esz = 48 * 2 * fe_per_coord
    assert len(raw) == 1000 * esz, len(raw)
    shape = (1000, 2, 6) if fe_per_coord == 1 else (1000, 2, 2, 6)
    xy = np.zeros(shape, dtype=np.uint64)
    inf = np.zeros(1000, dtype=np.uint8)
    for k in range(1000):
        e = bytearray(raw[k * esz:(k + 1) * esz])
        flags = e[0] >> 5
        e[0] &= 0x1F
        assert (flags & 0b100) == 0, "uncompressed file has compression flag"
        if flags & 0b010:
            inf[k] = 1
            assert all(b == 0 for b in e)
            continue
        fes = [int.from_bytes(e[i * 48:(i + 1) * 48], "big") for i in range(2 * fe_per_coord)]
        if fe_per_coord == 1:
            xy[k, 0] = limbs(fes[0])
            xy[k, 1] = limbs(fes[1])
        else:  # c1 first, then c0
            xy[k, 0, 1] = limbs(fes[0]); xy[k, 0, 0] = limbs(fes[1])
            xy[k, 1, 1] = limbs(fes[2]); xy[k, 1, 0] = limbs(fes[3])
    return xy, inf


def main():
    os.makedirs(OUT, exist_ok=True)
    t = os.path.join(REF, "bls12_381/src/curves/tests")
    xy, inf = parse_multiples(os.path.join(t, "g1_uncompressed_valid_test_vectors.dat"), 1)
    np.savez_compressed(os.path.join(OUT, "bls12_381_g1_multiples.npz"), xy=xy, infinity=inf)
    xy, inf = parse_multiples(os.path.join(t, "g2_uncompressed_valid_test_vectors.dat"), 2)
    np.savez_compressed(os.path.join(OUT, "bls12_381_g2_multiples.npz"), xy=xy, infinity=inf)

    pts = []
    for f in sorted(glob.glob(os.path.join(REF, "bls12_377/src/curves/tests/*G2*.json"))):
        j = json.load(open(f))
        for v in j["vectors"]:
            for key in ("P", "Q0", "Q1"):
                if key not in v:
                    continue
                c = []
                for coord in ("x", "y"):
                    c0, c1 = [int(s, 16) for s in v[key][coord].split(",")]
                    c.append([limbs(c0), limbs(c1)])
                pts.append(c)
    arr = np.array(pts, dtype=np.uint64)
    np.savez_compressed(os.path.join(OUT, "bls12_377_g2_h2c_points.npz"), xy=arr)
    print("g1/g2 multiples + %d BLS12-377 G2 points written to %s" % (len(pts), OUT))
    # item lists of the two traits the Rust binding implements (tests/test_rust_boundary.py checks the binding against
    # them where the reference tree is not available)
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), ""))
    sys.path.insert(0, os.path.dirname(OUT))
    import test_rust_boundary as T
    with open(T.GOLDEN, "w") as f:
        json.dump(T._reference_items(), f, indent=1)
    print("trait item lists written to %s" % T.GOLDEN)


if __name__ == "__main__":
    main()
