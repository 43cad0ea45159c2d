"""CPU oracle for the DriveSceneGen denoising hot path -- TEST INFRASTRUCTURE ONLY.

This package is a torch-CPU restatement of the arithmetic the reference executes
through un-vendored third-party code (diffusers==0.20.0, accelerate==0.22.0,
torch==2.0.1; /root/reference/requirements.txt:4,15,16).  It is NOT part of the
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
this path and diffusers/torchvision cannot be imported in
this image, so the oracle is pinned only by (i) the structural / integer /
constant known-answer values of SURVEY.md Appendix C and (ii) its own committed
golden vectors under tests/golden/.
"""
