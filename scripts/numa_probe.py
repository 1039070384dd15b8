"""Where does the process run relative to the GPU (dev tool)?  Prints the GPU's NUMA node (sysfs), the CPUs of that
node, and the CPU the main thread is on."""
import glob, os, sys
import torch
p = torch.cuda.get_device_properties(0)
bus = getattr(p, "pci_bus_id", None)
print("props:", p.name, "pci_bus_id", bus, "pci_device_id", getattr(p, "pci_device_id", None), "domain", getattr(p, "pci_domain_id", None))
for d in glob.glob("/sys/class/drm/card*/device"):
    try:
        print(d, "numa_node", open(d + "/numa_node").read().strip(), "local_cpulist", open(d + "/local_cpulist").read().strip(),
              "vendor", open(d + "/vendor").read().strip())
    except OSError as e:
        print(d, e)
print("main thread on cpu", os.sched_getaffinity(0).__len__(), "allowed;", "current cpu:", open("/proc/self/stat").read().split()[38])
for n in glob.glob("/sys/devices/system/node/node*/cpulist"):
    print(n, open(n).read().strip())
