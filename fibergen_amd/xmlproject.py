"""Project XML tree with fibergen's path syntax.

Mirrors FGProject::get_path / get / set / erase (F:26632-26736):
  * paths are relative to <settings>, components separated by '.';
  * '..' introduces an attribute:  'solver..n'  -> attribute n of <solver>;
  * 'name[i]' (or 'name(i)') picks the i-th child called name;
  * set() creates missing elements (as many as needed to reach index i).
"""
from __future__ import annotations

import re
import xml.etree.ElementTree as ET

_ATTR = "<xmlattr>"


class XMLPathError(RuntimeError):
    pass


def _split(path):
    full = "settings." + path
    full = full.replace("..", "." + _ATTR + ".")
    return full.split(".")


def _name_index(part):
    elems = re.split(r"[\[\]()]", part)
    name = elems[0]
    index = int(elems[1]) if len(elems) > 1 and elems[1] != "" else 0
    return name, index


class XMLProject:
    def __init__(self):
        self.reset()

    def reset(self):
        self.root = ET.Element("settings")

    # -- loading / saving ------------------------------------------------
    def set_xml(self, text):
        root = ET.fromstring(text)
        if root.tag != "settings":
            # boost property_tree keeps whatever the document root is; everything is read
            # below 'settings', so a different root simply yields an empty project
            wrapper = ET.Element("settings")
            root = wrapper
        self.root = root

    def load_xml(self, filename):
        with open(filename, "r") as f:
            self.set_xml(f.read())

    def get_xml(self):
        return '<?xml version="1.0" encoding="utf-8"?>\n' + ET.tostring(self.root, encoding="unicode")

    # -- path access -------------------------------------------------------
    def _walk(self, path, create):
        """Returns (element, attribute_name or None).  create: 1 create, 0 must exist, -1 erase."""
        parts = _split(path)
        assert parts[0] == "settings"
        cur = self.root
        i = 1
        while i < len(parts):
            part = parts[i]
            last = i == len(parts) - 1
            if part == _ATTR:
                if i + 1 >= len(parts):
                    raise XMLPathError("XML path '%s' not found" % path)
                attr = parts[i + 1]
                if i + 2 < len(parts):
                    raise XMLPathError("XML path '%s' not found" % path)
                if create < 0:
                    cur.attrib.pop(attr, None)
                    return None, None
                if create == 0 and attr not in cur.attrib:
                    raise XMLPathError("XML path '%s' not found" % path)
                return cur, attr
            if part == "":
                # trailing '.' as in set('solver.materials.fiber.', E=10): stay on the element
                i += 1
                continue
            name, index = _name_index(part)
            same = [c for c in cur if c.tag == name]
            if index < len(same):
                nxt = same[index]
                if create < 0 and last:
                    cur.remove(nxt)
                    return None, None
            else:
                if create > 0:
                    nxt = None
                    for _ in range(len(same), index + 1):
                        nxt = ET.SubElement(cur, name)
                elif create < 0:
                    return None, None
                else:
                    raise XMLPathError("XML path '%s' not found" % path)
            cur = nxt
            i += 1
        return cur, None

    def get(self, path):
        el, attr = self._walk(path, 0)
        if attr is not None:
            return el.attrib[attr]
        return (el.text or "")

    def set(self, path, value=""):
        el, attr = self._walk(path, 1)
        if attr is not None:
            el.set(attr, value)
        else:
            el.text = value

    def erase(self, path):
        self._walk(path, -1)

    # -- tree helpers used by the action interpreter ---------------------------
    def child(self, *names):
        cur = self.root
        for n in names:
            if cur is None:
                return None
            cur = cur.find(n)
        return cur
