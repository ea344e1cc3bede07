import os, sys, runpy
from lightcurver_amd import _lib
if os.environ.get('LCMI_ALT_LIB'):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ['LCMI_ALT_LIB'])
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
