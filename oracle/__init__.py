"""CPU oracle for the Stereo R-CNN inference hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  The product path (stereo_rcnn_amd/)
never imports it and fails loudly when its HIP library is missing.
Parity status: PINNED against outputs of the reference itself, produced in the build container (the reference ships
no tests or golden vectors of its own - SURVEY.md 8(c)):
  * its Python (network, stereo RPN, proposal layer, anchors, box transforms, ROI level routing, heads, box_estimator
    cost/gradient closures and solutions, infer_boundary, KITTI writer, dense_align / Box3d) is imported from
    /root/reference/lib under the shims of tests/golden/reference_shims.py and run on seeded inputs; the outputs are
    committed as tests/golden/reference_*.npz (generator: tests/golden/make_reference_golden.py) and the oracle
    reproduces them bit for bit (tests/test_reference_golden.py);
  * its two CUDA kernels (NMS, ROIAlign) are compiled UNCHANGED for gfx950 (oracle/build.py:build_ref ->
    oracle/_ref/libref_ops*.so) and executed on the MI355X against the C restatement and the product kernels
    (tests/test_ref_kernels_gpu.py): bit-equal.
  * demo.py's decode and per-class NMS blocks (script code) are sliced out of the file and exec'd on the reference
    network's outputs; postprocess.py gives the same numbers exactly.
  * OpenCV's float INTER_LINEAR resize of the preprocessing (third-party arithmetic; opencv-python is unpinned in the
    reference's requirements.txt and absent here) is restated from the published algorithm (oracle/preprocess.py cites
    resize.cpp) and pinned to hand-computed known answers + a second loop-level transcription
    (tests/test_preprocess_cpu.py) -- not to outputs of cv2 itself, which cannot be produced in this image.
"""
