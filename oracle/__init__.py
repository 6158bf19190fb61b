"""CPU oracle for the DeepLabv3+ head training step.

TEST INFRASTRUCTURE ONLY.  This package restates, in plain PyTorch / numpy, the
arithmetic of the reference's ASPP, Decoder, SynchronizedBatchNorm, cross-entropy
and Evaluator so that the HIP path can be checked against it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the shipped package never does.

Parity pin: every function here is compared, in ``tests/test_oracle_golden.py``,
with fixtures under ``tests/golden/`` that ``tools/gen_fixtures.py`` produced by
importing the reference itself (torch 2.10.0, numpy 2.2.6, CPU).
"""
