"""Build-owned stand-in for the MathWorks `matlab` Python package, used ONLY by
tests/golden/gen_golden.py so that the reference's modules import in a container
without MATLAB.  `matlab.double(x)` is only ever used by the reference to
marshal lists for the engine call; identity keeps the values inspectable."""


def double(x):
    return x
