"""Child process of tests/test_gpu_parity.py::test_layer_launches_across_the_tag_wrap: CFX_LIBCFX_PATH points at libcfx_dev.so."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

if __name__ == "__main__":
    import test_gpu_parity as T
    name, cid, N, C = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    T.tagwrap_case(name, cid, N, C)
    print("ok")
