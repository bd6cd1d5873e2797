"""pip install -e .   /   python setup.py build_hip

Builds pymgrid_amd/libmgx.so in-tree with hipcc for gfx950 (the same recipe as __graft_entry__.build(): mgx_abi.hip + the
five slices of mgx_fused.hip compiled in parallel, pymgrid_amd/_lib.py) and packages it with the Python surface.  hipcc cross-compiles without a GPU."""
import importlib.util
import os

from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

ROOT = os.path.dirname(os.path.abspath(__file__))


def build_libmgx():
    spec = importlib.util.spec_from_file_location("_mgx_lib_build", os.path.join(ROOT, "pymgrid_amd", "_lib.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                      # ctypes / subprocess only: no torch import needed to compile
    return mod.build(verbose=True)


class BuildHip(Command):
    description = "compile pymgrid_amd/libmgx.so with hipcc (--offload-arch=gfx950)"
    user_options = []

    def initialize_options(self):
        pass

    def finalize_options(self):
        pass

    def run(self):
        print("libmgx:", build_libmgx())


class BuildPy(build_py):
    def run(self):
        build_libmgx()
        super().run()


class Develop(develop):
    def run(self):
        build_libmgx()
        super().run()


setup(
    name="pymgrid_amd",
    version="0.1.0",
    description="MI355X-native batched microgrid-step engine (HIP kernels behind a C ABI and a pymgrid.envs-shaped surface)",
    packages=find_packages(include=["pymgrid_amd", "pymgrid_amd.*"]),
    package_data={"pymgrid_amd": ["libmgx.so", "data/*.npz", "csrc/*"]},
    data_files=[("include", ["include/mgx.h"])],
    python_requires=">=3.10",
    install_requires=["numpy", "torch"],
    cmdclass={"build_hip": BuildHip, "build_py": BuildPy, "develop": Develop},
)
