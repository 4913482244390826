#!/usr/bin/env python3
"""Writes tests/golden/circuit{1,2}_matrices.json: the reference's own golden constraint matrices for this path --
the ONLY known-answer vectors /root/reference holds -- transcribed from its test sources:

  circuit1  /root/reference/relations/src/gr1cs/tests/circuit1.rs:28-61   (checked by tests/mod.rs:78-103)
  circuit2  /root/reference/relations/src/gr1cs/tests/circuit2.rs:19-43   (checked by tests/mod.rs:136-147)

Format: {predicate label: [matrix, ...]}, a matrix = list of rows, a row = list of [coefficient, column] pairs
(`Matrix<F> = Vec<Vec<(F, usize)>>`, utils/matrix.rs:4).  The reference is Rust and cannot be imported or run in this
image, so the literals below were copied by hand from the two files above; the tests compare the oracle's
`to_matrices()` output with the JSON, not with anything inside oracle/.  Re-run after editing:  python tests/golden/make_r1cs_golden.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

CIRCUIT1 = {
    "R1CS": [[], [], []],
    "poly-predicate-A": [[[(1, 1)]], [[(1, 2)]], [[(1, 3)]], [[(1, 9)]]],
    "poly-predicate-B": [[[(1, 4)], [(1, 10)]], [[(1, 6)], [(1, 11)]], [[(1, 10)], [(1, 13)]]],
    "poly-predicate-C": [[[(1, 7)], [(1, 9), (1, 10)]], [[(1, 8)], [(1, 13)]], [[(1, 11)], [(1, 5)]]],
}

CIRCUIT2 = {"R1CS": [
    [[(1, 1)], [(1, 1)], [(1, 0)]],
    [[(2, 2)], [(1, 1), (1, 2)], [(2, 1), (2, 2)]],
    [[(1, 3)], [(1, 1), (1, 2)], [(2, 1), (2, 2)]],
]}


def main():
    for name, data in (("circuit1_matrices.json", CIRCUIT1), ("circuit2_matrices.json", CIRCUIT2)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
