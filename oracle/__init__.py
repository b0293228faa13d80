"""oracle/ — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under oracle/ is imported by the product path (pytorch-segmentation_amd/): only tests/,
__graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it, and only as the checker.

The reference (yassouali/pytorch-segmentation) is pure Python on torch's CPU operators, so the
restatement is functional torch-CPU fp32 code keyed by the reference's own state_dict names:

  oracle/weights.py       deterministic synthetic weights from a (key, shape) manifest
  oracle/pspnet_ref.py    PSPNet / dilated ResNet forward   (models/pspnet.py, models/resnet.py)
  oracle/losses_ref.py    CrossEntropy / Dice / Focal / Lovasz-Softmax  (utils/losses.py, utils/lovasz_losses.py)
  oracle/reference_harness.py + gen_golden.py
                          import the REAL reference from /root/reference (build container only) and
                          write tests/golden/*.pt; the GPU box never reads /root/reference.

Pinning: the reference ships no tests or golden vectors, so the oracle is pinned
against outputs of the reference code itself run in the build container (tests/golden/, generated
by oracle/gen_golden.py); tests/test_oracle_golden.py checks every restatement against them.
"""
