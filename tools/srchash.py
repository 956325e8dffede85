"""Hash of the kernel sources, to tie a committed counter profile to the code it was taken from."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "rust-raytracer_amd", "csrc", "*")))
    files.append(os.path.join(ROOT, "rust-raytracer_amd", "Makefile"))  # the compiler flags shape the code too
    for f in files:
        if f.endswith((".hip", ".hpp", ".h", ".cpp", "Makefile")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
