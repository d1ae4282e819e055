def connect_matlab():
    raise RuntimeError("no MATLAB in this container")
