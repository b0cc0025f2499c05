#!/bin/bash
# before every gpurun: the product library must be rebuilt from the current sources (built .so files travel, sources are not compiled there)
set -e
cd "$(dirname "$0")/.."
python -m eeg_image_decode_amd.build > /dev/null
python -m pytest tests/test_host_api.py -q -x 2>&1 | tail -1
