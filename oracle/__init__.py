"""CPU oracle for the MixerMDM denoising hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU (fp32) / numpy (float64 schedule tables)
restatement of the reference algorithm on the path SURVEY.md section 8(a)
lists.  It is the checker, never the product:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import it;
  * nothing under ``mixermdm_amd/`` imports it, and the product path raises if
    the HIP library is missing instead of falling back to this code.

Parity pinning: every function cites the reference file:line it follows and is
checked against golden vectors captured by importing the reference itself in
the builder container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_golden.py``).  The reference ships no tests or fixtures of
its own, so those captured vectors are the pin.

Third-party arithmetic: ``torch`` CPU kernels (matmul, softmax, erf, atan2, ...),
the same library the reference executes on.  One function is PARITY UNPINNED:
``oracle.encoder.clip_text_tower`` (and ``clip_encode_text``) restates the published
architecture of OpenAI CLIP's text transformer (package ``clip==1.0``, environment.yaml:49;
call sites src/models/mixermdm.py:212-217, 297-303), which is not under /root/reference
and cannot be imported here; everything around the tower in the text stage is pinned
(tests/golden/text.npz).
"""
