import os, glob, torch
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if 'pci' in k})
print('affinity', len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], '...')
for d in glob.glob('/sys/class/drm/card*/device'):
    try:
        print(d, open(d + '/numa_node').read().strip(), open(d + '/local_cpulist').read().strip(), os.path.basename(os.path.realpath(d)))
    except Exception as e:
        print(d, e)
print(open('/proc/self/status').read().split('Cpus_allowed_list')[1].split('\n')[0])
for n in sorted(glob.glob('/sys/devices/system/node/node*/cpulist')):
    print(n, open(n).read().strip())
