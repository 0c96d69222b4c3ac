"""A stand-in for ``mpi4py.MPI.Comm`` across REAL processes (mpi4py / mpirun are not in this image): the pickle-based
collectives the package's ``MpiBootstrap`` uses -- ``Get_rank``, ``Get_size``, ``allgather``, ``alltoall``, ``bcast``, ``barrier`` -- over files
in a shared directory (every collective: each rank writes ``<seq>.<rank>`` atomically, then reads everybody's).  Same method names
and semantics as mpi4py's lower-case object collectives; nothing of torch.distributed anywhere.  Test infrastructure only."""
import os
import pickle
import time


class FileComm:
    def __init__(self, directory, rank, size, timeout=120.0):
        self.dir, self.rank, self.size, self.timeout = directory, int(rank), int(size), float(timeout)
        self.seq = 0
        self.calls = {"allgather": 0, "alltoall": 0, "bcast": 0, "barrier": 0}

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def _exchange(self, obj):
        self.seq += 1
        tmp = os.path.join(self.dir, f".tmp.{self.seq}.{self.rank}")
        with open(tmp, "wb") as f:
            pickle.dump(obj, f)
        os.rename(tmp, os.path.join(self.dir, f"{self.seq}.{self.rank}"))
        out, t0 = [], time.time()
        for r in range(self.size):
            path = os.path.join(self.dir, f"{self.seq}.{r}")
            while not os.path.exists(path):
                if time.time() - t0 > self.timeout:
                    raise TimeoutError(f"FileComm: rank {r} never reached collective {self.seq}")
                time.sleep(0.002)
            with open(path, "rb") as f:
                out.append(pickle.load(f))
        return out

    def allgather(self, obj):
        self.calls["allgather"] += 1
        return self._exchange(obj)

    def alltoall(self, objs):
        self.calls["alltoall"] += 1
        assert len(objs) == self.size
        return [row[self.rank] for row in self._exchange(list(objs))]

    def bcast(self, obj, root=0):
        self.calls["bcast"] += 1
        return self._exchange(obj if self.rank == root else None)[root]

    def barrier(self):
        self.calls["barrier"] += 1
        self._exchange(None)

    Barrier = barrier
