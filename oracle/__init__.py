"""ORACLE - TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatements of the reference's SegmentClassifier forward
(reference gnn/model.py:14-156), used only as the checker:

* `oracle.dense_torch`  - the reference's own dense-incidence `bmm` formulation,
  statement for statement, on torch CPU tensors (same ATen ops as the reference).
* `oracle.index_numpy`  - index-form (gather / scatter-add) restatement in numpy.
* `oracle.index_c`      - the same index form in plain C (`oracle/index_c/segclf_oracle.c`,
  built by `oracle/Makefile` into `oracle/_build/`), fp32 and fp64-accumulate variants;
  the fast checker for large graphs and one leg of bench.py's `cpu_baseline`.

Pinning: the reference ships no golden vectors, tests or checkpoints for this path
(SURVEY.md section 4), so the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF:
`oracle/gen_golden.py` imports `/root/reference/gnn/model.py` unmodified in the build
container, runs it on seeded inputs and writes `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every restatement here against those files.

Only `tests/`, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may import
this package.  Nothing under `gnn-fpga_amd/` imports it; the product path has no CPU
fallback and raises when the HIP library is missing.
"""
