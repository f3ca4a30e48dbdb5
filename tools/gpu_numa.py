"""What the box says about where GPU 0 hangs: sysfs numa_node / local_cpulist of its PCI function."""
import glob, os
import torch
p = torch.cuda.get_device_properties(0)
bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
print("device 0", p.name, bdf)
for f in ("numa_node", "local_cpulist"):
    try:
        print(f, open(f"/sys/bus/pci/devices/{bdf}/{f}").read().strip())
    except OSError as e:
        print(f, "unreadable:", e)
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n, open(n).read().strip())
print("visible env", {k: v for k, v in os.environ.items() if "VISIBLE" in k})
