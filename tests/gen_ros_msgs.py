"""Generates C++ message / service structs from the reference's .msg / .srv DATA files (never its sources) into a
temporary include directory, for the API-conformance syntax check of tests/test_wrapper_sources.py."""
import os
import re

PRIM = {"bool": "uint8_t", "uint8": "uint8_t", "int8": "int8_t", "uint16": "uint16_t", "int16": "int16_t",
        "uint32": "uint32_t", "int32": "int32_t", "uint64": "uint64_t", "int64": "int64_t", "float32": "float",
        "float64": "double", "string": "std::string"}


def _fields(text, pkg):
    consts, fields, includes = [], [], set()
    for line in text.splitlines():
        line = line.split("#")[0].strip()
        if not line:
            continue
        m = re.match(r"(\S+)\s+(\w+)\s*=\s*(\S+)$", line)
        if m:
            consts.append("  enum { %s = %s };" % (m.group(2), m.group(3)))
            continue
        typ, name = line.split()[:2]
        arr = typ.endswith("[]")
        typ = typ[:-2] if arr else typ
        if typ in PRIM:
            cpp = PRIM[typ]
        elif typ in ("Header", "std_msgs/Header"):
            cpp = "std_msgs::Header"
            includes.add("std_msgs/Header.h")
        else:
            p, t = typ.split("/") if "/" in typ else (pkg, typ)
            cpp = "%s::%s" % (p, t)
            includes.add("%s/%s.h" % (p, "PoseGraph" if p == "pose_graph_tools_msgs" else t))
        scalar = not arr and typ in PRIM and typ != "string"
        fields.append("  %s %s%s;" % ("std::vector<%s>" % cpp if arr else cpp, name, " = 0" if scalar else ""))
    return consts, fields, includes


def generate(ref_root, out_dir, pkg="dpgo_ros"):
    os.makedirs(os.path.join(out_dir, pkg), exist_ok=True)
    for fn in sorted(os.listdir(os.path.join(ref_root, "msg"))):
        name = fn[:-4]
        consts, fields, inc = _fields(open(os.path.join(ref_root, "msg", fn)).read(), pkg)
        with open(os.path.join(out_dir, pkg, name + ".h"), "w") as f:
            f.write("#pragma once\n#include <cstdint>\n#include <memory>\n#include <string>\n#include <vector>\n")
            for i in sorted(inc):
                f.write("#include <%s>\n" % i)
            f.write("namespace %s {\nstruct %s {\n%s\n%s\n};\ntypedef std::shared_ptr<const %s> %sConstPtr;\n"
                    "typedef std::shared_ptr<%s> %sPtr;\n}\n"
                    % (pkg, name, "\n".join(consts), "\n".join(fields), name, name, name, name))
    for fn in sorted(os.listdir(os.path.join(ref_root, "srv"))):
        name = fn[:-4]
        req, res = open(os.path.join(ref_root, "srv", fn)).read().split("---")
        with open(os.path.join(out_dir, pkg, name + ".h"), "w") as f:
            f.write("#pragma once\n#include <cstdint>\n#include <vector>\n")
            parts, inc = [], set()
            for suffix, text in (("Request", req), ("Response", res)):
                c, fl, i = _fields(text, pkg)
                inc |= i
                parts.append("struct %s%s {\n%s\n%s\n};" % (name, suffix, "\n".join(c), "\n".join(fl)))
            for i in sorted(inc):
                f.write("#include <%s>\n" % i)
            f.write("namespace %s {\n%s\nstruct %s { typedef %sRequest Request; typedef %sResponse Response; "
                    "Request request; Response response; };\n}\n" % (pkg, "\n".join(parts), name, name, name))
