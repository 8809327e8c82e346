"""Throughput of the crop producer: 64 person boxes per launch out of 8 resident 640x480 pictures."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from rsgnet_amd import crops

rs = np.random.RandomState(0)
pics = [torch.from_numpy(rs.randint(0, 256, (480, 640, 3)).astype(np.uint8)).cuda() for _ in range(8)]
B = 64
which = [i % 8 for i in range(B)]
boxes = [[rs.uniform(0, 400), rs.uniform(0, 200), rs.uniform(60, 200), rs.uniform(120, 280)] for _ in range(B)]
cs = [crops.box_to_center_scale(b, 0.75) for b in boxes]
prod = crops.CropProducer((192, 256))
out = torch.empty(B, 3, 256, 192, device="cuda")
c, s = [a for a, _ in cs], [b for _, b in cs]
for _ in range(3):
    prod(pics, which, c, s, out=out)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
N = 50
for _ in range(N):
    prod(pics, which, c, s, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# kernel alone: reuse one descriptor table
from rsgnet_amd import hip
desc = (hip.CropDesc * B)()
for i in range(B):
    m = crops.get_affine_transform(c[i], s[i], 0.0, (192, 256))
    desc[i].image, desc[i].height, desc[i].width, desc[i].stride = pics[which[i]].data_ptr(), 480, 640, 1920
    for j, v in enumerate(m.reshape(6)):
        desc[i].m[j] = float(v)
table = torch.frombuffer(bytearray(bytes(desc)), dtype=torch.uint8).cuda()
st = torch.cuda.current_stream().cuda_stream
e0.record()
for _ in range(N):
    hip.lib().rsg_crop_warp_affine(table.data_ptr(), out.data_ptr(), None, B, 256, 192, prod._mean, prod._std, 1, st)
e1.record()
torch.cuda.synchronize()
k_us = e0.elapsed_time(e1) * 1e3 / N
print(f"crop producer: {B / dt:.0f} crops/s end to end (host maps + table upload + kernel, {dt * 1e6:.0f} us per batch); "
      f"kernel alone {k_us:.1f} us per batch = {B / k_us * 1e6:.0f} crops/s, "
      f"{B * 3 * 256 * 192 * 4 / k_us / 1e3:.0f} GB/s written")
