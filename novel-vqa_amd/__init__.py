"""novel-vqa on MI355X: HIP kernels + C ABI (csrc/, libnvqa.so) and the host-side
mirror of the reference's training-step interface (host/).  Import through
__graft_entry__.load_package() (the directory name is not a valid module name)."""
from .host import binding, dataset, h5, t7, trainer  # noqa: F401
