"""CPU oracle for the Stereo R-CNN inference hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  The product path (stereo_rcnn_amd/)
never imports it and fails loudly when its HIP library is missing.
Parity status: UNPINNED by the reference (no importable/buildable reference and
no upstream tests or golden vectors - see DESIGN.md).
"""
