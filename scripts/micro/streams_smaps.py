import torch, re
def big():
    out=[]; head=None; rss=0
    for l in open('/proc/self/smaps'):
        if re.match(r'^[0-9a-f]+-[0-9a-f]+ ', l):
            head=l.strip(); 
        elif l.startswith('Size:'):
            size=int(l.split()[1])
        elif l.startswith('Rss:'):
            rss=int(l.split()[1])
            if size>100000: out.append((size>>10, rss>>10, head[:40]))
    return out
torch.cuda.init()
x=torch.ones(10,device='cuda'); torch.cuda.synchronize()
print("after init:", big())
ss=[torch.cuda.Stream() for _ in range(4)]
for s in ss:
    with torch.cuda.stream(s):
        y=x+1
torch.cuda.synchronize()
print("after 4 streams used:", big())
