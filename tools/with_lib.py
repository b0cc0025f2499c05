"""A/B of two builds of the kernel library on one box:  python tools/with_lib.py <libeegclip_*.so> <script.py> [args...]
runs the script with the package bound to that library instead of the in-tree one (development aid; the product has no such switch)."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeg_image_decode_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name="__main__")
