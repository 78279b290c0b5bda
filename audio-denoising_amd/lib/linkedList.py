"""Doubly linked list of window slots used by the noise profiler (None marks a gap)."""


class Node:
    __slots__ = ("data", "prev", "next")

    def __init__(self, data, prev=None, next=None):
        self.data, self.prev, self.next = data, prev, next

    def _walk(self, attr):
        cur = getattr(self, attr)
        while cur is not None and cur.data is None:
            cur = getattr(cur, attr)
        return cur

    def getNextWithValidData(self):
        """Nearest following node that carries a window, or None."""
        return self._walk("next")

    def getPrevWithValidData(self):
        """Nearest preceding node that carries a window, or None."""
        return self._walk("prev")


class LinkedList:
    def __init__(self):
        self.first = self.last = None

    def append(self, data):
        node = Node(data, prev=self.last)
        if self.last is None:
            self.first = node
        else:
            self.last.next = node
        self.last = node

    def getAsList(self):
        out, cur = [], self.first
        while cur is not None:
            out.append(cur)
            cur = cur.next
        return out
