"""Message classes for the part of ReadServer's wire schema the service slice touches
(src/service/readserver.proto:3-14,31-37,39-49,56-64), re-typed here as a FileDescriptorProto so
the Python protobuf runtime can serialise golden bytes without protoc.  TEST INFRASTRUCTURE."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label, type_name=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    return f


def build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "readserver_slice.proto"
    fd.syntax = "proto2"
    req = fd.message_type.add()
    req.name = "Request"
    e = req.enum_type.add()
    e.name = "RequestType"
    for n, v in (("CountReads", 1), ("ExactMatch", 2), ("KmerMatch", 3), ("SiteMatch", 4)):
        x = e.value.add(); x.name, x.number = n, v
    e = req.enum_type.add()
    e.name = "ReturnType"
    for n, v in (("Count", 1), ("Reads", 2), ("All", 3), ("Samples", 4)):
        x = e.value.add(); x.name, x.number = n, v
    _field(req, "t", 1, _F.TYPE_ENUM, _F.LABEL_REQUIRED, ".Request.RequestType")
    _field(req, "rt", 2, _F.TYPE_ENUM, _F.LABEL_REQUIRED, ".Request.ReturnType")
    _field(req, "q", 3, _F.TYPE_STRING, _F.LABEL_REQUIRED)
    for n, num in (("k", 4), ("s", 5), ("p", 6)):
        _field(req, n, num, _F.TYPE_INT32, _F.LABEL_OPTIONAL)
    _field(req, "a", 7, _F.TYPE_STRING, _F.LABEL_OPTIONAL)
    _field(req, "isalt", 8, _F.TYPE_INT32, _F.LABEL_OPTIONAL)

    rc = fd.message_type.add()
    rc.name = "ResultCount"
    _field(rc, "c", 1, _F.TYPE_INT32, _F.LABEL_REQUIRED)
    rcount = fd.message_type.add()
    rcount.name = "ReplyCount"
    _field(rcount, "forward_matches", 1, _F.TYPE_MESSAGE, _F.LABEL_OPTIONAL, ".ResultCount")
    _field(rcount, "revcomp_matches", 2, _F.TYPE_MESSAGE, _F.LABEL_OPTIONAL, ".ResultCount")

    rr = fd.message_type.add()
    rr.name = "ResultReads"  # readserver.proto:35-37
    _field(rr, "r", 1, _F.TYPE_STRING, _F.LABEL_REQUIRED)
    rreads = fd.message_type.add()
    rreads.name = "ReplyReads"  # readserver.proto:61-64
    _field(rreads, "forward_matches", 1, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".ResultReads")
    _field(rreads, "revcomp_matches", 2, _F.TYPE_MESSAGE, _F.LABEL_REPEATED, ".ResultReads")

    rep = fd.message_type.add()
    rep.name = "Reply"
    e = rep.enum_type.add()
    e.name = "RequestType"
    for n, v in (("CountReads", 1), ("ExactMatch", 2), ("KmerMatch", 3), ("SiteMatch", 4)):
        x = e.value.add(); x.name, x.number = n, v
    e = rep.enum_type.add()
    e.name = "ReplyType"
    for n, v in (("ReplyCount", 1), ("ReplyReads", 2), ("ReplyAll", 3), ("ResultSamples", 4)):
        x = e.value.add(); x.name, x.number = n, v
    _field(rep, "rt", 1, _F.TYPE_ENUM, _F.LABEL_REQUIRED, ".Reply.RequestType")
    _field(rep, "t", 2, _F.TYPE_ENUM, _F.LABEL_REQUIRED, ".Reply.ReplyType")
    _field(rep, "q", 3, _F.TYPE_STRING, _F.LABEL_REQUIRED)
    _field(rep, "c", 4, _F.TYPE_MESSAGE, _F.LABEL_OPTIONAL, ".ReplyCount")
    _field(rep, "r", 5, _F.TYPE_MESSAGE, _F.LABEL_OPTIONAL, ".ReplyReads")

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName(n))
    return get("Request"), get("Reply")
