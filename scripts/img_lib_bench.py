"""kernel times of the image-gradient step for a given library build: img_lib_bench.py <lib.so> [random|smooth]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import depth_estimation_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "img_grad_bench.py")).read())
