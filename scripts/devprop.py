import torch
p=torch.cuda.get_device_properties(0)
print(p)
print("shared per block", getattr(p,'shared_memory_per_block',None), getattr(p,'shared_memory_per_block_optin',None), "per SM", getattr(p,'shared_memory_per_multiprocessor',None))
