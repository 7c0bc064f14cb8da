"""Minimal MJCF document model: just enough of the format to read the fruit-fly model
(`fruitfly/assets/fruitfly.xml` in the reference) and to apply the edits the reference walker
and tasks perform on it before compilation (`fruitfly/fruitfly.py:115-326`, `tasks/base.py`).

Supported: nested `<default class=...>` inheritance, `childclass` propagation through the body
tree, explicit `class=`, element removal, attribute edits. Everything is plain Python objects;
numeric attributes are parsed lazily by `Element.num()`.
"""

from __future__ import annotations

import xml.etree.ElementTree as ET
from typing import Iterator, Optional

import numpy as np

# All actuator shortcuts share one defaults slot in MJCF ("general").
ACTUATOR_TAGS = ("general", "motor", "position", "velocity", "cylinder", "muscle", "adhesion")


class Element:
    def __init__(self, tag: str, attrib: dict, parent: Optional["Element"] = None):
        self.tag = tag
        self.attrib = dict(attrib)
        self.parent = parent
        self.children: list[Element] = []

    # -- tree ---------------------------------------------------------------------------------
    def add(self, child: "Element") -> "Element":
        child.parent = self
        self.children.append(child)
        return child

    def remove(self) -> None:
        self.parent.children.remove(self)
        self.parent = None

    def iter(self, tag: Optional[str] = None) -> Iterator["Element"]:
        """Depth-first, document order (same order dm_control's `find_all` yields)."""
        for c in list(self.children):
            if tag is None or c.tag == tag:
                yield c
            yield from c.iter(tag)

    def child(self, tag: str) -> Optional["Element"]:
        for c in self.children:
            if c.tag == tag:
                return c
        return None

    @property
    def name(self) -> str:
        return self.attrib.get("name", "")

    # -- attributes ---------------------------------------------------------------------------
    def num(self, key: str, default=None) -> Optional[np.ndarray]:
        v = self.attrib.get(key)
        if v is None:
            return default
        if isinstance(v, str):
            return np.array([float(x) for x in v.split()], dtype=np.float64)
        return np.atleast_1d(np.asarray(v, dtype=np.float64))

    def set(self, key: str, value) -> None:
        self.attrib[key] = value

    def __repr__(self) -> str:
        return f"<{self.tag} {self.name!r}>"


def _convert(node: ET.Element, parent: Optional[Element]) -> Element:
    e = Element(node.tag, node.attrib, parent)
    for ch in node:
        e.children.append(_convert(ch, e))
    return e


class DefaultClass:
    def __init__(self, name: str, parent: Optional["DefaultClass"]):
        self.name = name
        self.parent = parent
        self.own: dict[str, dict] = {}  # tag -> attributes written at this level only

    def own_attr(self, tag: str, key: str):
        if tag in ACTUATOR_TAGS:
            for t in ACTUATOR_TAGS:
                if key in self.own.get(t, {}):
                    return self.own[t][key]
            return None
        return self.own.get(tag, {}).get(key)

    def resolved(self, tag: str) -> dict:
        out = self.parent.resolved(tag) if self.parent else {}
        if tag in ACTUATOR_TAGS:
            for t in ACTUATOR_TAGS:  # document order inside one <default> is not tracked; the
                out.update(self.own.get(t, {}))  # fly model never sets two shortcuts at once.
        else:
            out.update(self.own.get(tag, {}))
        return out


class Document:
    """A parsed MJCF file with its default-class table."""

    def __init__(self, path: str):
        self.path = path
        self.root = _convert(ET.parse(path).getroot(), None)
        self.classes: dict[str, DefaultClass] = {}
        top = self.root.child("default")
        self.main = DefaultClass("main", None)
        self.classes["main"] = self.main
        if top is not None:
            self._read_defaults(top, self.main)

    def _read_defaults(self, node: Element, cls: DefaultClass) -> None:
        for ch in node.children:
            if ch.tag == "default":
                sub = DefaultClass(ch.attrib["class"], cls)
                self.classes[sub.name] = sub
                self._read_defaults(ch, sub)
            else:
                cls.own.setdefault(ch.tag, {}).update(ch.attrib)

    # -- queries ------------------------------------------------------------------------------
    def section(self, tag: str) -> Optional[Element]:
        return self.root.child(tag)

    @property
    def worldbody(self) -> Element:
        return self.root.child("worldbody")

    def find(self, tag: str, name: str) -> Optional[Element]:
        if tag == "actuator":
            sec = self.section("actuator")
            for e in sec.children if sec else []:
                if e.name == name:
                    return e
            return None
        if tag == "tendon":
            sec = self.section("tendon")
            for e in sec.children if sec else []:
                if e.name == name:
                    return e
            return None
        if tag == "sensor":
            sec = self.section("sensor")
            for e in sec.children if sec else []:
                if e.name == name:
                    return e
            return None
        for e in self.root.iter(tag):
            if e.name == name and not self._in_defaults(e):
                return e
        return None

    def _in_defaults(self, e: Element) -> bool:
        p = e.parent
        while p is not None:
            if p.tag == "default":
                return True
            p = p.parent
        return False

    def find_all(self, tag: str) -> list[Element]:
        if tag in ("actuator", "tendon", "sensor"):
            sec = self.section(tag)
            return list(sec.children) if sec else []
        return [e for e in self.worldbody.iter(tag)]

    def class_of(self, e: Element) -> DefaultClass:
        """Effective default class: explicit `class`, else nearest ancestor body `childclass`."""
        c = e.attrib.get("class")
        if c is not None:
            return self.classes[c]
        p = e if e.tag == "body" else e.parent
        # a body's own childclass applies to its children, not to itself
        if e.tag == "body":
            p = e.parent
        while p is not None:
            cc = p.attrib.get("childclass")
            if cc is not None and p.tag in ("body", "worldbody", "frame"):
                return self.classes[cc]
            p = p.parent
        return self.main

    def resolved(self, e: Element) -> dict:
        """Element attributes with defaults folded in (element's own values win)."""
        out = self.class_of(e).resolved(e.tag)
        out.update(e.attrib)
        out.pop("class", None)
        return out

    def rnum(self, e: Element, key: str, default=None):
        v = self.resolved(e).get(key)
        if v is None:
            return default
        if isinstance(v, str):
            return np.array([float(x) for x in v.split()], dtype=np.float64)
        return np.atleast_1d(np.asarray(v, dtype=np.float64))
