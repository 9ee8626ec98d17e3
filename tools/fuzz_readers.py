"""Mutation fuzz of the .r1cs / witness readers against an AddressSanitizer + UBSan build of libligero_host.so:
    g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fPIC -shared -o build/asan/libligero_host.so ligero_amd/host/ligero_host.cpp
    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" \\
        ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=2048 python tools/fuzz_readers.py
(CPU only; sanitizers are not available on the GPU pool)"""
import ctypes, os, random, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "build", "asan", "libligero_host.so"))
L.lgh_last_error.restype = ctypes.c_char_p
L.lgh_circuit_from_r1cs.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_char_p]
L.lgh_circuit_destroy.argtypes = [ctypes.c_void_p]
L.lgh_read_witness.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
G = os.path.join(ROOT, "tests", "golden")
rng = random.Random(1)
files = {n: open(os.path.join(G, n), "rb").read() for n in ("cube.r1cs", "poseidon.r1cs", "poseidon_witness.wtns", "poseidon_witness.json", "multiplication.r1cs")}
d = tempfile.mkdtemp()
n_ok = n_err = 0
for it in range(int(os.environ.get("LG_FUZZ_ITERS", "3000"))):
    name = rng.choice(list(files))
    data = bytearray(files[name])
    mode = rng.randrange(4)
    if mode == 0:
        data = data[:rng.randrange(len(data) + 1)]
    elif mode == 1:
        for _ in range(rng.randrange(1, 8)):
            data[rng.randrange(len(data))] = rng.randrange(256)
    elif mode == 2:
        pos = rng.randrange(max(1, len(data) - 8)); data[pos:pos + 4] = (rng.choice([0, 1, 0xFFFFFFFF, 0x7FFFFFFF, 1 << 20])).to_bytes(4, "little")
    else:
        pos = rng.randrange(max(1, len(data) - 12)); data[pos:pos + 8] = (rng.choice([0, 1 << 40, (1 << 64) - 1, 12345])).to_bytes(8, "little")
    path = os.path.join(d, "f")
    open(path, "wb").write(bytes(data))
    if name.endswith(".r1cs"):
        h = ctypes.c_void_p()
        rc = L.lgh_circuit_from_r1cs(ctypes.byref(h), path.encode())
        if rc == 0:
            n_ok += 1; L.lgh_circuit_destroy(h)
        else:
            n_err += 1
    else:
        cnt = ctypes.c_uint64(0)
        buf = (ctypes.c_uint64 * (4 * 300))()
        rc = L.lgh_read_witness(path.encode(), buf, 300, ctypes.byref(cnt))
        n_ok += rc == 0; n_err += rc != 0
print("fuzz done: ok", n_ok, "rejected", n_err)
