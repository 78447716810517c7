"""Print VGPR/SGPR/scratch/occupancy/LDS per kernel from `make -C gnn-fpga_amd/csrc asm` remarks."""
import re
import subprocess
import sys

for f in sys.argv[1:] or ["build/gnn_kernels.remarks", "build/sell_pipeline.remarks"]:
    txt = open(f).read()
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        name = b.split()[0]
        g = lambda k: re.search(k + r": (\d+)", b).group(1)
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(anonymous namespace\)::", "", dn).split("(")[0]
        print(dn.ljust(40), "VGPR", g("VGPRs").rjust(3), "SGPR", g("TotalSGPRs").rjust(3), "scratch",
              g(r"ScratchSize \[bytes/lane\]").rjust(4), "occ", g(r"Occupancy \[waves/SIMD\]"),
              "LDS", g(r"LDS Size \[bytes/block\]"))
