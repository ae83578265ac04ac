"""CPU oracle for the single-stage detection inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mydetection_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

What it is: a restatement, in plain torch-CPU / numpy / C, of the reference
algorithm (duanzhiihao/myDetection) for
    backbone convs -> FPN -> head -> box decode -> conf filter/top-k -> class-aware NMS
with every function citing the reference file:line it follows.

How it is pinned: ``oracle/gen_golden.py`` imports the real reference from
/root/reference in the build container (third-party imports it lacks are stubbed)
and writes inputs+outputs to ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this restatement against those vectors.

PARITY UNPINNED at one boundary: ``torchvision.ops.nms`` (called at
utils/structures.py:133,162) is a third-party native op that is neither vendored
in the reference nor installed here, and the reference holds no tests or golden
vectors for it.  ``oracle/nms_ref.c`` / ``oracle.postprocess.nms_single_class``
restate torchvision's published CPU kernel (torchvision/csrc/ops/cpu/nms_kernel.cpp,
`nms_kernel_impl`): areas=(x2-x1)*(y2-y1); stable descending sort by score; greedy
pass suppressing j when inter/(area_i+area_j-inter) > thr, float32 arithmetic,
comparison against the threshold as double; kept indices in score order.
"""
