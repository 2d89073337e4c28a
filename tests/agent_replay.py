"""The statements of the reference's ``Agent`` that touch the flow and its optimizer, replayed one by one for the checkpoint tests.

agent.py itself cannot be imported (tensorboard, unconditional ``.cuda()``), so the tests replay the statements of
  * ``Agent.__init__``  agent.py:20-28  (get_flow -> DataParallel -> optim.Adam(self.flow.parameters(), lr) -> MultiStepLR)
  * ``Agent.save_ckpt`` agent.py:132-152 (``self.flow.module.cpu().state_dict()``, ``self.optimizer_flow.state_dict()``, torch.save, ``.cuda()``)
  * ``Agent.load_ckpt`` agent.py:171-198 (torch.load to CPU, ``self.flow.module.load_state_dict``, ``self.flow.cuda()``, a NEW
    ``optim.Adam(self.flow.parameters(), lr)``, ``load_state_dict`` of the saved optimizer state, clock)
  * ``Agent.train_func`` agent.py:75-92 for the unconditional recipe
in that order, with ``device`` standing in for ``.cuda()`` ("cpu" on the build container: ``.cpu()`` / ``.cuda()`` are then no-ops on the
same objects, exactly the aliasing the real calls have).  ``get_flow`` is a parameter: this repo's, or the reference's own.
"""
import torch
import torch.optim as optim
from torch.nn import DataParallel


class ReplayedAgent:
    def __init__(self, config, get_flow, device, lr=1e-3, lr_decay=(1000,), gamma=0.1):
        self.config, self.device, self.lr = config, torch.device(device), lr
        self.flow = get_flow(config)                                             # agent.py:20
        self.flow = DataParallel(self.flow)                                      # agent.py:21
        self.optimizer_flow = optim.Adam(self.flow.parameters(), lr)             # agent.py:22-23
        self.scheduler = optim.lr_scheduler.MultiStepLR(self.optimizer_flow, milestones=list(lr_decay), gamma=gamma)     # agent.py:25-28
        self.clock = {"epoch": 0, "minibatch": 0, "iteration": 0}

    def train_func(self, rotation):
        """agent.py:75-92 (unconditional, uniform base): forward, loss, zero_grad, backward, step."""
        flow = self.flow.to(self.device)                                         # agent.py:52: self.flow.cuda()
        flow.train()
        _, ldjs = flow(rotation.to(self.device), None)
        loss = (-ldjs).mean()
        self.optimizer_flow.zero_grad()
        loss.backward()
        self.optimizer_flow.step()
        self.clock["minibatch"] += 1
        self.clock["iteration"] += 1
        return float(loss.detach())

    def save_ckpt(self, path):
        flow_state_dict = self.flow.module.cpu().state_dict()                    # agent.py:133
        save_dict = {                                                            # agent.py:139-145
            "clock": dict(self.clock, scheduler_0=self.scheduler.state_dict()),
            "flow_state_dict": flow_state_dict,
            "optimizer_flow_state_dict": self.optimizer_flow.state_dict(),
        }
        torch.save(save_dict, path)                                              # agent.py:151
        self.flow.to(self.device)                                                # agent.py:152: self.flow.cuda()

    def load_ckpt(self, path):
        checkpoint = torch.load(path, map_location=torch.device("cpu"), weights_only=False)      # agent.py:171
        self.flow.module.load_state_dict(checkpoint["flow_state_dict"])          # agent.py:190
        self.flow.to(self.device)                                                # agent.py:192: self.flow.cuda()
        self.optimizer_flow = optim.Adam(self.flow.parameters(), self.lr)        # agent.py:193-194
        self.optimizer_flow.load_state_dict(checkpoint["optimizer_flow_state_dict"])             # agent.py:195-197
        for k in ("epoch", "minibatch", "iteration"):                            # agent.py:198 (utils/utils.py:47-50)
            self.clock[k] = checkpoint["clock"][k]
        return checkpoint
