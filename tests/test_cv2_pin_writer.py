"""The recorder behind tests/test_cv2_optional.py (which needs a cv2 and is skipped in this image): what it writes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cv2_pin import Pin  # noqa: E402


def test_pin_file_layout(tmp_path):
    pin = Pin('4.5.1', str(tmp_path / 'profiles' / 'cv2_pin.json'))
    assert pin.call('cv.remap', True, '0 of 10 elements differ')
    assert not pin.call('cv.cvtColor HSV2RGB_FULL', False, 'max 2 LSB')
    pin.induced('HSV2RGB_FULL_over_2^24_cube', bytes_differing=12, bytes=50331648, rate=12 / 50331648, max_lsb=1)
    rec = json.load(open(pin.path))
    assert rec['cv2_version'] == '4.5.1' and [c['pass'] for c in rec['calls']] == [True, False]
    assert rec['summary'] == {'calls': 2, 'passed': 1, 'verdict': 'differences: see calls'}
    assert rec['induced']['HSV2RGB_FULL_over_2^24_cube']['bytes_differing'] == 12
