#!/bin/bash
# One recorded attempt to put a real OpenCV next to the oracle on the GPU box.  Nothing is installed into the
# image: `pip download` fetches a wheel into /tmp when an index is reachable, the wheel is unpacked under /tmp/cv and used through
# PYTHONPATH for tests/test_cv2_optional.py + tests/cv2_pin.py only.  Without a network the log of the attempt IS the result
# (profiles/cv2_install_attempt.txt): a recorded blocker instead of an assumption.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/cv2_install_attempt.txt
mkdir -p $ROOT/gpurun_out
{
  echo "== $(date -u +%FT%TZ) host $(hostname) =="
  echo "-- python -c 'import cv2'"
  python -c "import cv2; print('cv2', cv2.__version__, cv2.__file__)" 2>&1 | tail -1
  echo "-- any OpenCV on the box (find / -iname '*opencv*' -o -name 'cv2*')"
  find / \( -iname "*opencv*" -o -name "cv2*" \) -not -path "/proc/*" -not -path "$ROOT/*" -not -path "/root/repo/*" 2>/dev/null | head -20
  echo "-- local wheelhouse"
  ls /opt/wheelhouse 2>/dev/null | grep -i -E "opencv|cv2" || echo "(no opencv wheel in /opt/wheelhouse)"
  echo "-- name resolution / route"
  getent hosts pypi.org files.pythonhosted.org 2>&1 || echo "(pypi.org does not resolve)"
  (ip route 2>/dev/null || cat /proc/net/route) | head -5
  for spec in "opencv-python-headless==4.5.1.48" "opencv-python-headless==4.5.4.58" "opencv-python-headless>=4.5.4.58,!=4.7.0.68"; do
    echo "-- pip download --no-deps --dest /tmp/cvwheel '$spec'"
    timeout 90 python -m pip download --no-deps --disable-pip-version-check --retries 1 --timeout 10 --dest /tmp/cvwheel "$spec" 2>&1 | tail -8
    ls /tmp/cvwheel/*.whl 2>/dev/null && break
  done
  whl=$(ls /tmp/cvwheel/*.whl 2>/dev/null | head -1)
  if [ -n "$whl" ]; then
    echo "-- got $whl: unpacking under /tmp/cv (not installed)"
    mkdir -p /tmp/cv && cd /tmp/cv && python -m zipfile -e "$whl" . && cd $ROOT
    PYTHONPATH=/tmp/cv python -c "import cv2; print('cv2', cv2.__version__)"
    PYTHONPATH=/tmp/cv timeout 900 python -m pytest tests/test_cv2_optional.py -q 2>&1 | tail -15
    PYTHONPATH=/tmp/cv timeout 900 python tests/cv2_pin.py 2>&1 | tail -15
    cp -f profiles/cv2_pin.json gpurun_out/ 2>/dev/null
  else
    echo "== RESULT: no OpenCV wheel obtainable on the GPU box (no index reachable, none on disk): the cv2 boundary stays unpinned =="
  fi
} > $OUT 2>&1
tail -5 $OUT
