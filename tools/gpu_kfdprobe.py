"""What the kernel driver says about compute queues per process (/sys/class/kfd/kfd/proc/<pid>/queues/<n>/{gpuid,type,size}): the layout
api.hip: kfd_compute_queues reads for its census of a device's queues."""
import os, torch, glob
streams=[torch.cuda.Stream() for _ in range(6)]
for s in streams:
    with torch.cuda.stream(s): torch.zeros(4,device="cuda").add_(1)
torch.cuda.synchronize()
base="/sys/class/kfd/kfd/proc"
print("me", os.getpid(), os.listdir(base) if os.path.isdir(base) else "no proc dir")
for p in glob.glob(base+"/*"):
    try:
        print(p, os.listdir(p))
        q=os.path.join(p,"queues")
        if os.path.isdir(q):
            qs=os.listdir(q); print(" queues:", len(qs), qs[:4])
            for x in qs[:2]:
                for f in os.listdir(os.path.join(q,x)):
                    try: print("   ",x,f,open(os.path.join(q,x,f)).read().strip())
                    except Exception as e: print("   ",x,f,"ERR",e)
    except Exception as e:
        print(p,"ERR",e)
