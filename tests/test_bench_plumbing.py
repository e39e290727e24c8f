"""bench.py's host logic that needs no GPU: the legs of the default run derive their arguments (and their `metric`) from the
headline's instead of hard-coding them (round 2's records all carried the headline's metric string)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench


def _args(**kw):
    base = dict(backbone='ResNet50FPN', batch=8, height=800, width=1280, dtype='bf16', rotated_bbox=False, unit_rotation=False,
                steps=50, warmup=10, cpu_seconds=20.0, no_eager_leg=False, other_steps=20, mode='infer')
    base.update(kw)
    return argparse.Namespace(**base)


def test_short_names():
    assert bench.short_name('ResNet50FPN') == 'RN50FPN'
    assert bench.short_name('ResNet101FPN') == 'RN101FPN'
    assert bench.short_name('ResNeXt50_32x4dFPN') == 'RNX50_32x4dFPN'


def test_legs_override_only_what_they_name():
    head = _args()
    leg = bench.leg_args(head, backbone='ResNet101FPN', batch=16)
    assert (leg.backbone, leg.batch, leg.steps, leg.cpu_seconds, leg.no_eager_leg) == ('ResNet101FPN', 16, 20, 0.0, True)
    assert leg.warmup == 5 and leg.height == 800 and leg.dtype == 'bf16'
    assert (head.backbone, head.batch, head.steps, head.cpu_seconds) == ('ResNet50FPN', 8, 50, 20.0)      # the headline's own arguments are untouched
    rot = bench.leg_args(head, rotated_bbox=True, unit_rotation=True)
    assert rot.rotated_bbox and rot.unit_rotation and not head.rotated_bbox


def test_spec_constants_match_the_survey():
    # sparse-realistic = 23 224 of 11.52 M scores of P3 clear the threshold
    assert abs(bench.SPEC_FRACTION - 23224 / 11520000.0) < 1e-12
    assert bench.SPEC_CANDIDATES == [23224, 5646, 1456, 391, 93]
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.MFMA_BF16_PEAK_TFLOPS == 2500.0
