"""Per-launch table of conv_gemm_kernel from a rocprofv3 kernel trace of `bench.py` (B = 1024)."""
import csv, sys
def ops(B=1024):
    enc=[]
    def e(M,N,K): enc.append((M,N,K))
    e(B*100,64,49)
    for _ in range(2): e(B*25,64,576); e(B*25,64,576)
    e(B*9,128,576); e(B*9,128,64); e(B*9,128,1152); e(B*9,128,1152); e(B*9,128,1152)
    e(B*4,256,1152); e(B*4,256,128); e(B*4,256,2304); e(B*4,256,2304); e(B*4,256,2304)
    e(B,512,2304); e(B,512,256); e(B,512,4608); e(B,512,4608); e(B,512,4608)
    e(B,400,512)
    un=[]
    def u(L,N,K): un.append((B*L,N,K))
    def crb(L,cin,cout):
        u(L,cout,3*cin)
        if cin!=cout: u(L,cout,cin)
        u(L,cout,3*cout)
    u(64,512,6); u(64,512,2); u(64,512,1536)
    crb(64,512,512); u(32,512,1536)
    crb(32,512,1024); crb(32,1024,1024); u(16,1024,3072)
    crb(16,1024,2048); crb(16,2048,2048); crb(16,2048,2048); crb(16,2048,2048)
    crb(16,4096,1024); crb(16,1024,1024); u(16,1024,2048); u(16,1024,2048)
    crb(32,2048,512); crb(32,512,512); u(32,512,1024); u(32,512,1024)
    u(64,512,1536)
    return enc+[(B,28672,663)]+un
def main(path, call=5):
    rows=[r for r in csv.DictReader(open(path)) if 'conv_gemm' in r['Kernel_Name']]
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    o=ops(); per=len(o); c=rows[per*call:per*(call+1)]
    tot=tt=0
    out=[]
    for (M,N,K),r in zip(o,c):
        d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; fl=2*M*N*K; tot+=fl; tt+=d
        out.append((M,N,K,d,fl/d/1e6))
    return out, tot/tt/1e6, tt
if __name__=="__main__":
    tabs=[main(p) for p in sys.argv[1:]]
    for i,row in enumerate(tabs[0][0]):
        print(f"M={row[0]:6d} N={row[1]:5d} K={row[2]:5d} | "+" | ".join(f"{t[0][i][3]:7.1f}us {t[0][i][4]:7.1f}TF" for t in tabs))
    print("total:", " | ".join(f"{t[2]:8.1f}us {t[1]:7.1f}TF" for t in tabs))
