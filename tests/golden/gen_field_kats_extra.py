"""How the two round-6 entries of field_kats.json were made (values only; nothing of oracle/ or nova_amd/ is imported):
  mle_multi_evaluate_known_values  the literal inputs and outputs of the reference's test (src/spartan/polys/multilinear.rs:456-485)
  spmv_mixed_coefficients          the reference's fixture (src/r1cs/sparse.rs:486-520) and, because that test stores no output (it compares
                                   the reference's two implementations), the dense product over the INTEGERS: out[r] = sum_c M[r][c] z[c];
                                   a test reduces it mod p per field (a negative entry -k is the field element p - k)."""
ENTRIES = [(0, 0, 1), (0, 1, -1), (0, 2, 42), (1, 0, 2), (1, 1, 3), (1, 2, 4), (1, 3, 5), (1, 4, 6), (1, 5, 7), (2, 0, -2), (2, 1, -3), (2, 2, -7),
           (4, 0, 1), (4, 1, 3), (4, 2, -5), (4, 3, -1), (4, 4, 100)]


def dense(z, rows=5):
    out = [0] * rows
    for r, c, v in ENTRIES:
        out[r] += v * z[c]
    return out


if __name__ == "__main__":
    print(dense([1, 2, 3, 4, 5, 6]), dense([11, 12, 13, 14, 15, 16]))
