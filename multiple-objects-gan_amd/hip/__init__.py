"""ctypes binding (lib.py) and autograd wrappers (ops.py) over libmogan_hip.so."""
