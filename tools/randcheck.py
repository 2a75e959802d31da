import ctypes as C, os, sys
ROOT="/root/repo"
libc=C.CDLL(None)
libc.srand(1)
a=[libc.rand() for _ in range(3)]
libc.srand(1)
h=C.CDLL(os.path.join(ROOT,"travellingsalesmanoptimization_amd/host/libtsphost.so"))
h.tsp_gpu.restype=C.c_void_p
g=h.tsp_gpu()
b=[libc.rand() for _ in range(3)]
print("before gpu init:",a); print("after gpu init :",b, "ctx",g)
