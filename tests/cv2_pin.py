"""The record tests/test_cv2_optional.py leaves behind: profiles/cv2_pin.json -- one pass / fail line per cv2 call the oracle
restates, and the two induced counts (1/32-px map entries that differ between the closed-form cell homographies and
cv.getPerspectiveTransform(DECOMP_SVD); bytes of HSV2RGB_FULL over the 2^24 cube that differ from the cv2 build).  Anyone with
``opencv-python-headless`` in the reference's range turns "parity unpinned at the cv2 boundary" into evidence with
``pytest tests/test_cv2_optional.py`` and attaches that file."""
import json
import os
import platform
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, 'profiles', 'cv2_pin.json')


class Pin:

    def __init__(self, cv_version, path=PATH):
        self.path = path
        self.record = {'cv2_version': cv_version, 'python': platform.python_version(), 'machine': platform.machine(),
                       'written': time.strftime('%Y-%m-%d %H:%M:%S'), 'calls': [], 'induced': {}}

    def call(self, name, ok, detail=''):
        """One comparison oracle <-> cv2: recorded whether it holds or not (the test asserts on it afterwards)."""
        self.record['calls'].append({'call': name, 'pass': bool(ok), 'detail': str(detail)})
        self.write()
        return bool(ok)

    def induced(self, key, **numbers):
        self.record['induced'][key] = numbers
        self.write()

    def write(self):
        passed = sum(c['pass'] for c in self.record['calls'])
        self.record['summary'] = {'calls': len(self.record['calls']), 'passed': passed,
                                  'verdict': 'pinned' if passed == len(self.record['calls']) else 'differences: see calls'}
        os.makedirs(os.path.dirname(self.path), exist_ok=True)
        with open(self.path, 'w') as fout:
            json.dump(self.record, fout, indent=1)
